// HIP execution-model EMULATOR shim -- TEST INFRASTRUCTURE ONLY.
//
// `tests/emu/` lets the CPU test tier (no GPU in the build container) compile the UNMODIFIED
// kernel sources of gp_mpc_amd/csrc/ with g++ (this directory is put first on the include path,
// so `#include <hip/hip_runtime.h>` resolves here) and execute them with the GPU's semantics:
// one cooperative fiber per GPU thread, workgroups run one after another, `__syncthreads()` and
// the wave64 collectives (shuffles, readlane, the f64 MFMA with the gfx950 fragment layout) are
// rendez-vous points between fibers.  It exists to catch indexing / synchronisation / layout
// bugs before spending GPU minutes.  It is NOT a product path and NOT a fallback: the package
// loader (gp_mpc_amd/_lib.py) only ever loads the hipcc-built libgpmpc_hip.so and raises if it
// or the GPU is missing; the emulator library is built under tests/emu/_build and is handed to
// the host classes explicitly by tests/test_emu_*.py.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <functional>

#define GPMPC_EMULATED 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
// double ext-vectors only (d4 accumulators): clang ext_vector_type(n) -> gcc vector_size(8n)
#define ext_vector_type(n) vector_size((n) * 8)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

extern dim3 threadIdx, blockIdx, blockDim, gridDim;

struct alignas(16) double2 { double x, y; };
struct alignas(32) double4 { double x, y, z, w; };
using std::max;
using std::min;
inline long long __double_as_longlong(double v) { long long b; std::memcpy(&b, &v, 8); return b; }
inline double __longlong_as_double(long long b) { double v; std::memcpy(&v, &b, 8); return v; }
inline int __double2loint(double v) { int64_t b; std::memcpy(&b, &v, 8); return (int)(uint32_t)(b & 0xffffffff); }
inline int __double2hiint(double v) { int64_t b; std::memcpy(&b, &v, 8); return (int)(uint32_t)((uint64_t)b >> 32); }
inline double __hiloint2double(int hi, int lo) {
    uint64_t b = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
    double v; std::memcpy(&v, &b, 8); return v;
}

// ---------------------------------------------------------------- host runtime API subset
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
struct emu_stream { int id; };
typedef emu_stream* hipStream_t;
struct emu_event { double t; long seq_recorded; long seq_done; };
typedef emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                     hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; char gcnArchName[256]; size_t totalGlobalMem; };
typedef struct emu_graph* hipGraph_t;
typedef struct emu_graphexec* hipGraphExec_t;

namespace emu {
// shmem == 0: workgroups run one after another (static __shared__ arrays are shared by construction).
// shmem  > 0: ALL workgroups run concurrently as fibers, each with its own dynamic-LDS buffer
//             (emu::dyn_smem()), so kernels that synchronise BETWEEN workgroups through global
//             memory (persistent task-queue kernels) can be executed and can dead-lock like on a GPU.
// Launches are ASYNCHRONOUS like on a GPU: they are queued on their stream (nullptr = stream 0) and
// executed at the next synchronisation point (stream/event/device synchronise, memcpy, memset, free).
// Streams are in-order queues; the head kernels of different streams run concurrently (fiber
// interleaving), events order work across streams.
void launch(dim3 grid, dim3 block, const std::function<void()>& body, size_t shmem = 0, hipStream_t stream = nullptr);
void drain();                                   // run everything that is queued
void record(hipEvent_t e, hipStream_t s);
void wait_event(hipStream_t s, hipEvent_t e);
void* dyn_smem();
void yield_thread();
void syncthreads();
double wave_xchg(double v, int src_lane);                 // value held by src_lane (all 64 lanes call)
void mfma_f64_16x16x4(double a, double b, const double* c, double* d);
double now_ms();
int lane();
}  // namespace emu

inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    std::memset(p, 0, sizeof(*p));
    std::snprintf(p->name, sizeof(p->name), "HIP emulator (CPU fibers)");
    std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "emu");
    p->multiProcessorCount = 1;
    return hipSuccess;
}
template <class T> inline hipError_t hipMalloc(T** p, size_t bytes) {
    void* q = nullptr;
    if (posix_memalign(&q, 256, bytes ? bytes : 256) != 0) return hipErrorOutOfMemory;
    *p = (T*)q;
    return hipSuccess;
}
inline hipError_t hipFree(void* p) { emu::drain(); std::free(p); return hipSuccess; }
enum { hipHostMallocDefault = 0 };
inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { *p = std::calloc(bytes ? bytes : 1, 1); return *p ? hipSuccess : (hipError_t)2; }
inline hipError_t hipHostFree(void* p) { emu::drain(); std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { emu::drain(); std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { emu::drain(); std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height,
                                   hipMemcpyKind, hipStream_t) {
    emu::drain();
    for (size_t r = 0; r < height; ++r) std::memcpy((char*)d + r * dpitch, (const char*)s + r * spitch, width);
    return hipSuccess;
}
inline hipError_t hipMemset(void* d, int v, size_t n) { emu::drain(); std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { emu::drain(); std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { static int next_id = 1; *s = new emu_stream{next_id++}; return hipSuccess; }
constexpr unsigned hipStreamDefault = 0;
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
inline hipError_t hipStreamDestroy(hipStream_t s) { emu::drain(); delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { emu::drain(); return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { emu::drain(); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = (size_t)64 << 30; *total_b = (size_t)64 << 30; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated HIP error"; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event{0.0}; return hipSuccess; }
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event{0.0, 0, 0}; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { emu::wait_event(s, e); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { emu::record(e, s); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { emu::drain(); return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { emu::drain(); *ms = (float)(b->t - a->t); return hipSuccess; }

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    ::emu::launch(dim3(grid), dim3(block), [=]() { kern(__VA_ARGS__); }, (size_t)(shmem), (hipStream_t)(stream))

// ---------------------------------------------------------------- device-side subset
inline void __syncthreads() { emu::syncthreads(); }
inline double __shfl(double v, int src) { return emu::wave_xchg(v, src & 63); }
inline int __shfl(int v, int src) { return (int)emu::wave_xchg((double)v, src & 63); }
inline double __shfl_xor(double v, int mask) { return emu::wave_xchg(v, (emu::lane() ^ mask) & 63); }
inline double __shfl_down(double v, int delta) {
    int s = emu::lane() + delta;
    return emu::wave_xchg(v, s < 64 ? s : emu::lane());
}
inline int emu_readlane(int v, int src) { return (int)emu::wave_xchg((double)v, src); }
#define __builtin_amdgcn_readlane(v, l) emu_readlane((v), (l))
#define __builtin_amdgcn_readfirstlane(v) emu_readlane((v), 0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ::emu::yield_thread()
// memory-model builtins: the emulator is sequentially consistent, fences are no-ops
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
template <class T> inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <class T> inline void __hip_atomic_store(T* p, T v, int, int) { *p = v; }
template <class T> inline T __hip_atomic_fetch_add(T* p, T v, int, int) { T o = *p; *p = o + v; return o; }
template <class T> inline bool __hip_atomic_compare_exchange_strong(T* p, T* expected, T desired, int, int, int) {
    if (*p == *expected) { *p = desired; return true; }
    *expected = *p;
    return false;
}
inline void __threadfence() {}
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
#define __builtin_amdgcn_sched_barrier(x) ((void)0)

typedef double emu_d4 __attribute__((vector_size(32)));
inline emu_d4 emu_mfma_f64(double a, double b, emu_d4 c) {
    double ci[4] = {c[0], c[1], c[2], c[3]}, di[4];
    emu::mfma_f64_16x16x4(a, b, ci, di);
    emu_d4 d = {di[0], di[1], di[2], di[3]};
    return d;
}
// (the last operand: for the f64 matrix instruction the BLGP field holds NEG modifiers -- bit 0 negates A, bit 1 B, bit 2 C)
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) \
    emu_mfma_f64(((z) & 1) ? -(a) : (a), ((z) & 2) ? -(b) : (b), ((z) & 4) ? emu_d4{-(c)[0], -(c)[1], -(c)[2], -(c)[3]} : (c))

inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
inline int atomicCAS(int* p, int cmp, int val) { int o = *p; if (o == cmp) *p = val; return o; }
inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
using std::isnan;
using std::isfinite;
