// Latency of the operations a dependent chain of one wavefront is made of, on the device at hand (gfx950): one workgroup of one
// wavefront, N iterations of a loop whose every iteration depends on the one before through exactly the operation named.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/chain_ops tools/microbench/chain_ops.hip && tools/microbench/chain_ops
// Prints ns per iteration (HIP events around the launch, launch overhead subtracted with an empty kernel) and, in brackets, cycles at
// the clock a chain of dependent 32-bit vector additions implies (4 cycles each on a 16-lane SIMD).  K7's edit step (DESIGN.md
// section 7c) is such a chain: cross-lane move -> address -> load -> funnel shift / count -> ballot -> wave maximum -> scalar
// bookkeeping.  Not part of the product; nothing loads it.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                            \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                         \
            exit(1);                                                                        \
        }                                                                                   \
    } while (0)

// every CU busy for a while: the clocks an idle device has gone down to come back up
__global__ __launch_bounds__(256) void k_spin(int n, float *out) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) a = a * b + 0.5f;
    }
    if (a == 123.f) out[0] = a;
}

__global__ __launch_bounds__(64) void k_empty(int n, uint32_t *out) {
    if (n < 0) out[threadIdx.x] = 1;
}

// 8 dependent v_add per iteration
__global__ __launch_bounds__(64) void k_valu(int n, uint32_t *out) {
    uint32_t x = threadIdx.x;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(x));
    }
    out[threadIdx.x] = x;
}

// 8 dependent s_add per iteration
__global__ __launch_bounds__(64) void k_salu(int n, uint32_t *out) {
    uint32_t s = (uint32_t)n;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("s_add_u32 %0, %0, %0" : "+s"(s) : : "scc");
    }
    out[threadIdx.x] = s;
}

// x = x of lane + 1 through the LDS crossbar (ds_bpermute_b32)
__global__ __launch_bounds__(64) void k_bpermute(int n, uint32_t *out) {
    int x = threadIdx.x;
    const int src = ((threadIdx.x + 1) & 63) * 4;
    for (int i = 0; i < n; i++) x = __builtin_amdgcn_ds_bpermute(src, x) + 1;
    out[threadIdx.x] = x;
}

// x = x of lane - 1 with a DPP wave shift (the move + 1 add)
__global__ __launch_bounds__(64) void k_dpp(int n, uint32_t *out) {
    int x = threadIdx.x;
    for (int i = 0; i < n; i++) x = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, false) + 1;
    out[threadIdx.x] = x;
}

// the six-step maximum over the wavefront as K7 does it (row shifts, row broadcasts), then the readlane of lane 63
__device__ __forceinline__ int wave_max_i32(int v) {
    int t;
    t = __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xe, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xc, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;
    return __builtin_amdgcn_readlane(v, 63);
}
__global__ __launch_bounds__(64) void k_wavemax(int n, uint32_t *out) {
    int x = threadIdx.x;
    for (int i = 0; i < n; i++) x = (wave_max_i32(x) & 63) + (int)threadIdx.x;
    out[threadIdx.x] = x;
}

// vector compare -> ballot in scalar registers -> find first -> back into a vector operand
__global__ __launch_bounds__(64) void k_ballot(int n, uint32_t *out) {
    int x = threadIdx.x;
    for (int i = 0; i < n; i++) {
        const unsigned long long b = __ballot(x > 31);
        const int f = __ffsll((long long)b);
        x = ((int)threadIdx.x + f + i) & 63;
    }
    out[threadIdx.x] = x;
}

// readlane -> scalar -> vector
__global__ __launch_bounds__(64) void k_readlane(int n, uint32_t *out) {
    int x = threadIdx.x;
    for (int i = 0; i < n; i++) x = __builtin_amdgcn_readlane(x, 17) + (int)threadIdx.x;
    out[threadIdx.x] = x;
}

// pointer chase through LDS, one word per lane
__global__ __launch_bounds__(64) void k_lds(int n, uint32_t *out) {
    __shared__ uint32_t t[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) t[i] = (uint32_t)((i * 7 + 13) & 1023);
    __syncthreads();
    uint32_t x = threadIdx.x;
    for (int i = 0; i < n; i++) x = t[x];
    out[threadIdx.x] = x;
}

// pointer chase through global memory: `words` words of table (the footprint decides which cache answers), every lane its own chain
__global__ __launch_bounds__(64) void k_global(int n, const uint32_t *__restrict__ t, uint32_t spread, uint32_t *out) {
    uint32_t x = threadIdx.x * spread;   // spread 0: every lane walks the same chain (one line per load, as K7's neighbours nearly do)
    for (int i = 0; i < n; i++) x = t[x];
    out[threadIdx.x] = x;
}

// the same with K7's access: five adjacent words from an unaligned word address, funnel-shifted, xor-ed with a second stream's and
// counted (fetch64 + the compare of snake64), the count feeding the next address
__global__ __launch_bounds__(64) void k_fetch64(int n, const uint32_t *__restrict__ t, uint32_t mask, uint32_t *out) {
    uint32_t x = threadIdx.x * 37u;
    for (int i = 0; i < n; i++) {
        const uint32_t *p = t + (x >> 4), *q = t + ((x + 4099u) >> 4);
        const uint32_t s = (x & 15u) * 2u, s2 = ((x + 4099u) & 15u) * 2u;
        uint32_t a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            a[j] = (uint32_t)((((uint64_t)p[j + 1] << 32) | p[j]) >> s);
            b[j] = (uint32_t)((((uint64_t)q[j + 1] << 32) | q[j]) >> s2);
        }
        const uint32_t d0 = a[0] ^ b[0], d1 = a[1] ^ b[1], d2 = a[2] ^ b[2], d3 = a[3] ^ b[3];
        const int m = d0 ? (__builtin_ctz(d0) >> 1) : d1 ? 16 + (__builtin_ctz(d1) >> 1) : d2 ? 32 + (__builtin_ctz(d2) >> 1) : d3 ? 48 + (__builtin_ctz(d3) >> 1) : 64;
        x = (x + (uint32_t)m + 1u) & mask;
    }
    out[threadIdx.x] = x;
}

// scalar pointer chase (s_load_dword: the address is the same in every lane)
__global__ __launch_bounds__(64) void k_sload(int n, const uint32_t *__restrict__ t, uint32_t *out) {
    uint32_t x = 0;
    for (int i = 0; i < n; i++) x = t[__builtin_amdgcn_readfirstlane(x)];
    out[threadIdx.x] = x;
}

static double run(const char *name, void (*launch)(int), int n, double empty_ms, double per = 1.0, double ns_per_cycle = 0.0) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch(n / 10);  // warm-up
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, 0));
        launch(n);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double ns = (best - empty_ms) * 1e6 / n / per;
    if (name) {
        if (ns_per_cycle > 0) printf("%-58s %8.1f ns  [%6.0f cycles]\n", name, ns, ns / ns_per_cycle);
        else printf("%-58s %8.1f ns\n", name, ns);
    }
    return ns;
}

static uint32_t *g_out;
static uint32_t *g_tab;
static uint32_t g_mask;

int main() {
    CHECK(hipMalloc(&g_out, 64 * 4));
    const int n = 200000;
    {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_spin, dim3(4096), dim3(256), 0, 0, 2000000, (float *)g_out);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("warm-up: every CU busy for %.0f ms (%.1f TFLOP/s fp32 FMA)\n", ms, 4096.0 * 256 * 2000000 * 16 * 2 / (ms * 1e-3) / 1e12);
        }
    }
    const double empty = [] {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        double best = 1e30;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, 0, g_out);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        return best;
    }();
    printf("empty launch %.1f us\n", empty * 1e3);
    const double valu = run(nullptr, [](int k) { hipLaunchKernelGGL(k_valu, dim3(1), dim3(64), 0, 0, k, g_out); }, n, empty, 8.0);
    const double cyc = valu / 4.0;  // a dependent 32-bit vector add issues every 4 cycles at best; what this wavefront sees may be more
    printf("dependent v_add_u32                                        %8.2f ns  (taken as 4 cycles: %.3f ns per cycle, %.2f GHz)\n", valu, cyc, 1.0 / cyc);
    run("dependent s_add_u32", [](int k) { hipLaunchKernelGGL(k_salu, dim3(1), dim3(64), 0, 0, k, g_out); }, n, empty, 8.0, cyc);
    run("DPP wave shift + add", [](int k) { hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, k, g_out); }, n, empty, 1.0, cyc);
    run("ds_bpermute_b32 + add", [](int k) { hipLaunchKernelGGL(k_bpermute, dim3(1), dim3(64), 0, 0, k, g_out); }, n, empty, 1.0, cyc);
    run("readlane -> vector add", [](int k) { hipLaunchKernelGGL(k_readlane, dim3(1), dim3(64), 0, 0, k, g_out); }, n, empty, 1.0, cyc);
    run("compare -> ballot -> find first -> vector", [](int k) { hipLaunchKernelGGL(k_ballot, dim3(1), dim3(64), 0, 0, k, g_out); }, n, empty, 1.0, cyc);
    run("wave maximum (6 DPP steps) + readlane", [](int k) { hipLaunchKernelGGL(k_wavemax, dim3(1), dim3(64), 0, 0, k, g_out); }, n, empty, 1.0, cyc);
    run("LDS pointer chase (ds_read_b32)", [](int k) { hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, k, g_out); }, n, empty, 1.0, cyc);
    struct Foot { const char *name, *name64; size_t words; int iters; };
    const Foot feet[] = {{"global pointer chase, 8 KB table (vector L1), one line", "   ... 64 lanes, 64 chains", 2048, n},
                         {"global pointer chase, 1 MB table (L2), one line", "   ... 64 lanes, 64 chains", 1u << 18, n},
                         {"global pointer chase, 64 MB table (Infinity Cache), one line", "   ... 64 lanes, 64 chains", 1u << 24, n / 4},
                         {"global pointer chase, 1 GB table (HBM), one line", "   ... 64 lanes, 64 chains", 1u << 28, n / 8}};
    run("dependent v_add_u32 again (x 4)", [](int k) { hipLaunchKernelGGL(k_valu, dim3(1), dim3(64), 0, 0, k, g_out); }, n, empty, 2.0, cyc);
    for (const Foot &f : feet) {
        std::vector<uint32_t> h(f.words);
        // a random cyclic walk with a stride that defeats the line: next = (i * odd + c) mod words
        const uint64_t mul = 2654435761ull | 1ull;
        for (size_t i = 0; i < f.words; i++) h[i] = (uint32_t)((i * mul + 12345u) & (f.words - 1));
        CHECK(hipMalloc(&g_tab, f.words * 4));
        CHECK(hipMemcpy(g_tab, h.data(), f.words * 4, hipMemcpyHostToDevice));
        run(f.name, [](int k) { hipLaunchKernelGGL(k_global, dim3(1), dim3(64), 0, 0, k, g_tab, 0u, g_out); }, f.iters, empty, 1.0, cyc);
        run(f.name64, [](int k) { hipLaunchKernelGGL(k_global, dim3(1), dim3(64), 0, 0, k, g_tab, 16u, g_out); }, f.iters, empty, 1.0, cyc);
        if (f.words == 2048) run("scalar pointer chase, 8 KB table (s_load_dword)", [](int k) { hipLaunchKernelGGL(k_sload, dim3(1), dim3(64), 0, 0, k, g_tab, g_out); }, n, empty, 1.0, cyc);
        if (f.words == 2048 || f.words == (1u << 18)) {
            g_mask = (uint32_t)(f.words * 8 - 1);  // base positions in the table's first half: the five words stay inside it
            run(f.words == 2048 ? "K7's fetch64 x 2 + compare + count, 8 KB (vector L1)" : "K7's fetch64 x 2 + compare + count, 1 MB (L2)",
                [](int k) { hipLaunchKernelGGL(k_fetch64, dim3(1), dim3(64), 0, 0, k, g_tab, g_mask, g_out); }, n, empty, 1.0, cyc);
        }
        CHECK(hipFree(g_tab));
        run("dependent v_add_u32 again (x 4)", [](int k) { hipLaunchKernelGGL(k_valu, dim3(1), dim3(64), 0, 0, k, g_out); }, n, empty, 2.0, cyc);
    }
    return 0;
}
