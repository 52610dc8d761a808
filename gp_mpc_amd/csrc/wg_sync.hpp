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

#ifndef GPMPC_EMULATED
__device__ int g_poll_mode = 0;     // experiment switch (GPMPC_POLL): 1 = poll with an atomic RMW, 2 = system scope
#endif
__device__ __forceinline__ int flag_load(const int* p) {
#ifndef GPMPC_EMULATED
    if (g_poll_mode == 1) return __hip_atomic_fetch_add(const_cast<int*>(p), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (g_poll_mode == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_store(int* p, int v) {
#ifndef GPMPC_EMULATED
    if (g_poll_mode >= 2) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
#endif
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Flag store that other XCDs are guaranteed to see soon.  Measured on MI355X (r01, tile-owner workers): the
// sc1 store of a relaxed agent-scope atomic store is NOT written through -- it can sit dirty in the writer's
// XCD L2 until some later L2 write-back on that XCD, and when every workgroup of that XCD is polling, there
// is none: 239 workers polled a word for 0.3 s that the 240th had "stored" (and could read back itself).
// Uncached allocation (hipExtMallocWithFlags) and system scope made no difference.  Hence a second
// write-back right after the store.
__device__ __forceinline__ void flag_post(int* p, int v) {
    flag_store(p, v);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

// called by ALL threads of the workgroup after their plain global stores
__device__ __forceinline__ void wg_publish(int* flag, int value) {
    GPMPC_DRAIN_VM();
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        GPMPC_DRAIN_VM();
        flag_post(flag, value);
    }
}

// called by ALL threads: wait until *f0 >= v0 (and *f1 >= v1 if f1); false on time-out / global error.
// `slot` is an int in LDS used to broadcast the outcome.
__device__ __forceinline__ bool wg_wait2(const int* f0, int v0, const int* f1, int v1, int* err, int limit, int* slot,
                                         int errcode = 2) {
    if (threadIdx.x == 0) {
        int ok = 1, spins = 0;
        while (flag_load(f0) < v0 || (f1 && flag_load(f1) < v1)) {
            __builtin_amdgcn_s_sleep(8);
#ifndef GPMPC_EMULATED
            if (g_poll_mode == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
            if (flag_load(err) != 0) { ok = 0; break; }
            if (++spins > limit) {                 // errcode tells the host who gave up, err[-..] what it last saw
                const int a = flag_load(f0) >= v0, b = !f1 || flag_load(f1) >= v1;   // which one is missing
                flag_store(err, errcode + 100000000 * a + 200000000 * b);
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
