// TEST INFRASTRUCTURE ONLY -- never part of the product, never loaded by nextdenovo_amd/.
//
// A lane-accurate interpreter for the repo's HIP kernels, so that kernel LOGIC can be checked against the oracle on a
// machine without a GPU (`-m "not gpu"` tests).  tests/simt/build_simt.py compiles the unmodified kernel sources under
// nextdenovo_amd/csrc with g++ and this header standing in for <hip/hip_runtime.h>:
//   * every lane of a workgroup is a fibre (ucontext) with its own stack and registers; a workgroup runs on one host
//     thread, workgroups of a launch are spread over host threads;
//   * wave-wide operations (__ballot, __shfl*, readlane, DPP moves, wave_barrier) are rendezvous points of the 64 lanes of
//     a wavefront, __syncthreads() of the whole workgroup -- a lane that reads LDS another lane has written without such a
//     point in between sees stale data here (on hardware it would depend on lock-step execution), so missing barriers show
//     up as parity failures;
//   * `__shared__` is storage shared by the fibres of the workgroup (static thread_local);
//   * the runtime API (hipMalloc, streams, events, copies) is the host heap, executed synchronously.
// It says nothing about performance and is orders of magnitude slower than a CPU port would be: it exists to execute the
// same source the GPU executes.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define SIMT_EMULATION 1

// ------------------------------------------------------------------ language
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __constant__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct ulonglong2 { unsigned long long x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }

namespace simt {

struct Block;
struct Lane {
    dim3 tidx;
    unsigned tid, wave, lane;
    Block *blk;
};
struct BlockIds {
    dim3 bidx, bdim, gdim;
};
extern thread_local Lane *g_lane;
const BlockIds &block_ids();

struct Snap {
    const uint64_t *v;   // value every lane of the wavefront deposited
    uint64_t mask;       // lanes that took part
};
Snap wave_sync(uint64_t v);  // rendezvous of the live lanes of the calling lane's wavefront
void block_sync();           // rendezvous of the live lanes of the workgroup
void launch(const char *kernel_name, dim3 grid, dim3 block, size_t dynamic_lds, const std::function<void()> &body);
void *dynamic_lds();         // the workgroup's `extern __shared__` array (build_simt.py rewrites the declaration to a call of this)

template <class T> inline uint64_t bits(T v) {
    static_assert(sizeof(T) <= 8, "wave exchange of a type wider than 64 bits");
    uint64_t b = 0;
    memcpy(&b, &v, sizeof(T));
    return b;
}
template <class T> inline T unbits(uint64_t b) {
    T v;
    memcpy(&v, &b, sizeof(T));
    return v;
}

// source lane of a data-parallel-primitive move (CDNA ISA, "DPP_CTRL"); -1 = no source (the lane keeps `old`, or reads 0 with
// bound_ctrl)
inline int dpp_source(int lane, int ctrl) {
    const int row = lane & ~15, in_row = lane & 15;
    if (ctrl >= 0 && ctrl <= 0xff) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);  // quad_perm
    if (ctrl >= 0x101 && ctrl <= 0x10f) {                                                   // row_shl
        const int s = in_row + (ctrl & 15);
        return s < 16 ? row + s : -1;
    }
    if (ctrl >= 0x111 && ctrl <= 0x11f) {  // row_shr
        const int s = in_row - (ctrl & 15);
        return s >= 0 ? row + s : -1;
    }
    if (ctrl >= 0x121 && ctrl <= 0x12f) return row + ((in_row - (ctrl & 15)) & 15);  // row_ror
    if (ctrl == 0x130) return lane + 1 < 64 ? lane + 1 : -1;                           // wave_shl:1
    if (ctrl == 0x134) return (lane + 1) & 63;                                         // wave_rol:1
    if (ctrl == 0x138) return lane >= 1 ? lane - 1 : -1;                               // wave_shr:1
    if (ctrl == 0x13c) return (lane - 1) & 63;                                         // wave_ror:1
    if (ctrl == 0x140) return row + (15 - in_row);                                     // row_mirror
    if (ctrl == 0x141) return row + ((in_row & 8) | (7 - (in_row & 7)));               // row_half_mirror
    if (ctrl == 0x142) return row >= 16 ? row - 1 : -1;                                // row_bcast:15
    if (ctrl == 0x143) return row >= 32 ? 31 : -1;                                     // row_bcast:31
    abort();
}

template <class T> inline T dpp(T old, T src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const Snap s = wave_sync(bits(src));
    const int lane = (int)g_lane->lane;
    if (!((row_mask >> (lane >> 4)) & 1) || !((bank_mask >> ((lane & 15) >> 2)) & 1)) return old;
    const int from = dpp_source(lane, ctrl);
    if (from < 0 || !((s.mask >> from) & 1)) return bound_ctrl ? T(0) : old;
    return unbits<T>(s.v[from]);
}

}  // namespace simt

#define threadIdx (simt::g_lane->tidx)
#define blockIdx (simt::block_ids().bidx)
#define blockDim (simt::block_ids().bdim)
#define gridDim (simt::block_ids().gdim)
static const int warpSize = 64;

// ------------------------------------------------------------------ wave / workgroup operations
static inline void __syncthreads() { simt::block_sync(); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __builtin_amdgcn_wave_barrier() { (void)simt::wave_sync(0); }
static inline void __builtin_amdgcn_s_setprio(int) {}
// v_alignbit_b32: the low 32 bits of {hi, lo} >> (shift & 31)
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned shift) {
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (shift & 31u));
}

static inline unsigned long long __ballot(int pred) {
    const simt::Snap s = simt::wave_sync(pred ? 1u : 0u);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++)
        if (((s.mask >> i) & 1) && s.v[i]) m |= 1ull << i;
    return m;
}

template <class T> static inline T __shfl(T var, int src, int width = 64) {
    const simt::Snap s = simt::wave_sync(simt::bits(var));
    const int lane = (int)simt::g_lane->lane;
    const int from = (lane & ~(width - 1)) + (src & (width - 1));
    return simt::unbits<T>(s.v[from]);
}
template <class T> static inline T __shfl_up(T var, unsigned delta, int width = 64) {
    const simt::Snap s = simt::wave_sync(simt::bits(var));
    const int lane = (int)simt::g_lane->lane;
    const int from = lane - (int)delta;
    if (from < (lane & ~(width - 1))) return var;
    return simt::unbits<T>(s.v[from]);
}
template <class T> static inline T __shfl_down(T var, unsigned delta, int width = 64) {
    const simt::Snap s = simt::wave_sync(simt::bits(var));
    const int lane = (int)simt::g_lane->lane;
    const int from = lane + (int)delta;
    if (from >= (lane & ~(width - 1)) + width) return var;
    return simt::unbits<T>(s.v[from]);
}
template <class T> static inline T __shfl_xor(T var, int mask, int width = 64) {
    const simt::Snap s = simt::wave_sync(simt::bits(var));
    const int lane = (int)simt::g_lane->lane;
    const int from = lane ^ mask;
    if (from >= (lane & ~(width - 1)) + width) return var;
    return simt::unbits<T>(s.v[from]);
}
template <class T> static inline T __builtin_amdgcn_readlane(T var, int src) {
    const simt::Snap s = simt::wave_sync(simt::bits(var));
    return simt::unbits<T>(s.v[src & 63]);
}
template <class T> static inline T __builtin_amdgcn_readfirstlane(T var) {
    const simt::Snap s = simt::wave_sync(simt::bits(var));
    return simt::unbits<T>(s.v[__builtin_ctzll(s.mask)]);
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl) \
    simt::dpp((old), (src), (ctrl), (row_mask), (bank_mask), (bound_ctrl))

// ------------------------------------------------------------------ integer intrinsics, atomics
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

template <class T, class U> static inline T atomicAdd(T *p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U> static inline T atomicOr(T *p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U> static inline T atomicAnd(T *p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U> static inline T atomicExch(T *p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U, class V> static inline T atomicCAS(T *p, U expected, V desired) {
    T e = (T)expected;
    __atomic_compare_exchange_n(p, &e, (T)desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return e;   // the value found (== expected when the exchange happened)
}
template <class T, class U> static inline T atomicMax(T *p, U v) {
    T cur = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (cur < (T)v && !__atomic_compare_exchange_n(p, &cur, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return cur;
}
template <class T, class U> static inline T atomicMin(T *p, U v) {
    T cur = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (cur > (T)v && !__atomic_compare_exchange_n(p, &cur, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return cur;
}
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))

// ------------------------------------------------------------------ runtime API (host heap, synchronous)
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
struct simtStream;
struct simtEvent;
typedef simtStream *hipStream_t;
typedef simtEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventBlockingSync = 1, hipEventDisableTiming = 2 };
enum { hipDeviceScheduleAuto = 0, hipDeviceScheduleSpin = 1, hipDeviceScheduleYield = 2, hipDeviceScheduleBlockingSync = 4 };
enum { hipHostMallocDefault = 0 };
struct hipDeviceProp_t {
    char name[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
    char gcnArchName[256];
};

const char *hipGetErrorString(hipError_t);
const char *hipGetErrorName(hipError_t);
hipError_t hipGetLastError();
hipError_t hipGetDeviceCount(int *);
hipError_t hipSetDevice(int);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *, int);
hipError_t hipDeviceSynchronize();
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
hipError_t hipMalloc(void **, size_t);
hipError_t hipFree(void *);
hipError_t hipHostMalloc(void **, size_t, unsigned);
hipError_t hipHostFree(void *);
hipError_t hipMemcpy(void *, const void *, size_t, hipMemcpyKind);
hipError_t hipMemcpyAsync(void *, const void *, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipMemset(void *, int, size_t);
hipError_t hipMemsetAsync(void *, int, size_t, hipStream_t);
hipError_t hipStreamCreate(hipStream_t *);
hipError_t hipStreamCreateWithFlags(hipStream_t *, unsigned);
hipError_t hipStreamCreateWithPriority(hipStream_t *, unsigned, int);
hipError_t hipExtStreamCreateWithCUMask(hipStream_t *, unsigned, const unsigned *);
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned);
hipError_t hipEventCreate(hipEvent_t *);
hipError_t hipEventCreateWithFlags(hipEvent_t *, unsigned);
hipError_t hipSetDeviceFlags(unsigned);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventElapsedTime(float *, hipEvent_t, hipEvent_t);

namespace simt {
template <class F, class... A> inline void launch_v(const char *name, F f, dim3 grid, dim3 block, size_t shmem, hipStream_t, A... args) {
    launch(name, grid, block, shmem, [&]() { f(args...); });
}
}  // namespace simt
#define hipLaunchKernelGGL(kernel, ...) simt::launch_v(#kernel, [](auto... a_) { kernel(a_...); }, __VA_ARGS__)
