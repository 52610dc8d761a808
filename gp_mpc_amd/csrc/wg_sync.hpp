// Inter-workgroup hand-offs inside/between concurrently running kernels (gfx950: 8 XCDs with private,
// mutually non-coherent L2s; a CU's L1 is never refreshed by other CUs' stores).  Protocol of
// /opt/skills/guides/cdna_hip_programming.md Guideline 16, counter/flag form:
//   producer: every wave drains its stores -> barrier -> ONE lane: agent-scope release fence, drain again,
//             relaxed agent-scope flag store;
//   consumer: ONE lane polls the flag relaxed (bounded, with s_sleep), ONE agent-scope acquire fence,
//             barrier, then plain loads by everybody.
// Flags are monotone ints, zeroed by a hipMemsetAsync before every use.  Every spin is bounded; on a
// time-out the error word is set, all waiters give up and the host falls back to the barrier-free path.
#pragma once
#include <hip/hip_runtime.h>

namespace gpmpc {

#ifdef GPMPC_EMULATED
#define GPMPC_DRAIN_VM() ((void)0)
#else
#define GPMPC_DRAIN_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

__device__ __forceinline__ int flag_load(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_store(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// called by ALL threads of the workgroup after their plain global stores
__device__ __forceinline__ void wg_publish(int* flag, int value, int* flag2 = nullptr) {
    GPMPC_DRAIN_VM();
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        GPMPC_DRAIN_VM();
        flag_store(flag, value);
        if (flag2) flag_store(flag2, value);
    }
}

// Write-through form (guide G16 recipe R1): the payload leaves as 16-byte sc1 stores, which go through the XCD's L2
// to memory, so the publication needs no L2 write-back (buffer_wbl2, 1.7 us clean and several under load): every
// storing wave drains, barrier, ONE lane stores the flag.  The consumer side is unchanged.
#ifdef GPMPC_EMULATED
struct wt_rsrc_t { char* base; };
inline wt_rsrc_t wt_make_rsrc(void* base, unsigned) { return wt_rsrc_t{(char*)base}; }
inline void wt_store16(const wt_rsrc_t& r, unsigned off, double2 v) { *reinterpret_cast<double2*>(r.base + off) = v; }
#else
typedef __amdgpu_buffer_rsrc_t wt_rsrc_t;
__device__ __forceinline__ wt_rsrc_t wt_make_rsrc(void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void wt_store16(wt_rsrc_t r, unsigned off, double2 v) {
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), r, (int)off, 0, /*sc1*/ 16);
}
#endif
// called by ALL threads of the workgroup after their wt_store16 stores (and nothing else to publish)
__device__ __forceinline__ void wg_publish_wt(int* flag, int value, int* flag2 = nullptr) {
    GPMPC_DRAIN_VM();
    __syncthreads();
    if (threadIdx.x == 0) {
        flag_store(flag, value);
        if (flag2) flag_store(flag2, value);
    }
}

// called by ALL threads: wait until *f0 >= v0 (and *f1 >= v1 if f1); false on time-out / global error.
// `slot` is an int in LDS used to broadcast the outcome.
__device__ __forceinline__ bool wg_wait2(const int* f0, int v0, const int* f1, int v1, int* err, int limit, int* slot,
                                         int errcode = 2) {
    if (threadIdx.x == 0) {
        int ok = 1, spins = 0;
#ifdef GPMPC_POLL_SERIAL   // (r01-r04a: flag, sleep, error word -- two memory round trips per look)
        while (flag_load(f0) < v0 || (f1 && flag_load(f1) < v1)) {
            __builtin_amdgcn_s_sleep(8);
            if (flag_load(err) != 0) { ok = 0; break; }
#else
        for (;;) {
            // the flag(s) and the error word of one look travel together: ONE round trip (~1 us across XCDs) per look
            const int a = flag_load(f0), b = f1 ? flag_load(f1) : v1, e = flag_load(err);
            if (a >= v0 && b >= v1) break;
            if (e != 0) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(8);
#endif
#ifdef GPMPC_POLL_SERIAL
            if (++spins > limit) {                 // errcode tells the host who gave up, err[-..] what it last saw
#else
            if ((++spins >> 1) > limit) {          // (a look takes half the time: the give-up TIME stays what the host chose; no 2 * limit: it may be INT_MAX)
#endif
                const int ma = flag_load(f0) >= v0, mb = !f1 || flag_load(f1) >= v1;   // which one is missing
                flag_store(err, errcode + 100000000 * ma + 200000000 * mb);
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *slot = ok;
    }
    __syncthreads();
    const bool r = *slot != 0;
    __syncthreads();
    return r;
}

}  // namespace gpmpc
