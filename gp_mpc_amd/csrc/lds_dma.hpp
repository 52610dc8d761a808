// Global memory -> LDS loads through the DMA path of the load unit (buffer_load_dwordx4 ... lds) for gfx950:
// a wave instruction moves 64 x 16 bytes from per-lane source addresses to LDS address (wave-uniform base + 16 lane),
// without staging registers.  Completion is tracked by the wave's vmcnt; the data is visible to other waves after a
// counted s_waitcnt vmcnt in the issuing wave followed by a workgroup barrier.  The loads are inline assembly:
// with the compiler's own builtin, hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of every LDS read that follows
// one (checked in the assembly output), which would serialise any ring of LDS images.
// Users: gemm_f64_dma.hpp (operand slabs), chol_worker.hpp (panel blocks of the trailing update).
#pragma once
#include <hip/hip_runtime.h>

namespace gpmpc {

#ifdef GPMPC_EMULATED
struct dma_rsrc_t { const char* base; unsigned bytes; };
inline dma_rsrc_t dma_make_rsrc(const void* base, unsigned bytes) { return dma_rsrc_t{(const char*)base, bytes}; }
// wave-uniform LDS destination + 16 * lane
inline void dma_load16(const dma_rsrc_t& r, char* lds_wave_base, unsigned voff, unsigned soff) {
    const unsigned long o = (unsigned long)voff + soff;
    char* dst = lds_wave_base + 16 * (threadIdx.x & 63);
    if (o + 16 > r.bytes) { for (int i = 0; i < 16; ++i) dst[i] = 0; return; }
    for (int i = 0; i < 16; ++i) dst[i] = r.base[o + i];
}
inline void dma_load16_relaxed(const dma_rsrc_t& r, char* lds_wave_base, unsigned voff, unsigned soff) {
    dma_load16(r, lds_wave_base, voff, soff);
}
// (LDS addresses as integers: under the emulator "LDS" is host memory and the handle is the pointer itself)
typedef unsigned long lds_addr_t;
inline lds_addr_t lds_addr_of(void* p) { return (lds_addr_t)p; }
inline void dma_load16(const dma_rsrc_t& r, lds_addr_t lds_wave_base, unsigned voff, unsigned soff) {
    dma_load16(r, (char*)lds_wave_base, voff, soff);
}
inline void dma_load_block64(const dma_rsrc_t& r, char* lds_wave_base, unsigned voff, unsigned half) {
    for (int sl = 0; sl < 4; ++sl) {
        dma_load16(r, lds_wave_base + 8192 * sl, voff, 128u * sl);
        dma_load16(r, lds_wave_base + 8192 * sl + 4096, voff, 128u * sl + half);
    }
}
template <int N> inline void dma_wait() {}
inline void dma_barrier() { __syncthreads(); }
inline void lds_barrier() { __syncthreads(); }
inline void lds_flush() {}
#else
typedef int dma_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma_rsrc_t dma_make_rsrc(const void* base, unsigned bytes) {
    const unsigned long b = (unsigned long)base;
    dma_rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));   // stride 0, no swizzle
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void dma_load16(dma_rsrc_t r, char* lds_wave_base, unsigned voff, unsigned soff) {
    const unsigned dst = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)lds_wave_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(__builtin_amdgcn_readfirstlane(dst)), "v"(voff), "s"(r), "s"(soff)
                 : "memory");
}
// The same with the LDS address as an integer the caller formed ONCE (lds_addr_of): the conversion of a generic pointer to an
// LDS address carries a null check, four scalar instructions per load when it is repeated for every load of a ring (r05).
typedef unsigned lds_addr_t;
__device__ __forceinline__ lds_addr_t lds_addr_of(void* p) {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long)(__attribute__((address_space(3))) char*)p);
}
__device__ __forceinline__ void dma_load16(dma_rsrc_t r, lds_addr_t lds_wave_base, unsigned voff, unsigned soff) {
    // (M0 is declared clobbered instead of being saved and restored around every load: two scalar instructions less)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(__builtin_amdgcn_readfirstlane((int)lds_wave_base)), "v"(voff), "s"(r), "s"(soff)
                 : "memory", "m0");
}
// This wave's share of one 64 x 64 block as ONE instruction sequence (r05): the eight loads of dma_load16(r, base + 8192 sl
// [+ 4096], voff, 128 sl [+ half]), sl = 0 .. 3 -- four 16-column slab images x two row halves (chol_worker.hpp).  M0 and the
// two scalar offsets advance by s_add between the loads: 3 scalar instructions per load instead of the ~10 of eight separate
// dma_load16 calls (generic -> LDS address conversion with its null check, M0 saved and restored around every load) -- the
// requesting wave of a SIMD reaches its own matrix instructions ~100 scalar instructions earlier per tile.
// (One instruction sits between every write of M0 and the load that reads it, as in dma_load16.)
__device__ __forceinline__ void dma_load_block64(dma_rsrc_t r, char* lds_wave_base, unsigned voff, unsigned half) {
    const unsigned dst = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)lds_wave_base;
    unsigned keep, s0, s1;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_mov_b32 %1, 0\n\t"
        "s_mov_b32 %2, %6\n\t"
        "buffer_load_dwordx4 %4, %5, %1 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %4, %5, %2 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_addk_i32 %1, 0x80\n\t"
        "buffer_load_dwordx4 %4, %5, %1 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_addk_i32 %2, 0x80\n\t"
        "buffer_load_dwordx4 %4, %5, %2 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_addk_i32 %1, 0x80\n\t"
        "buffer_load_dwordx4 %4, %5, %1 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_addk_i32 %2, 0x80\n\t"
        "buffer_load_dwordx4 %4, %5, %2 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_addk_i32 %1, 0x80\n\t"
        "buffer_load_dwordx4 %4, %5, %1 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_addk_i32 %2, 0x80\n\t"
        "buffer_load_dwordx4 %4, %5, %2 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(s0), "=&s"(s1)
        : "s"(__builtin_amdgcn_readfirstlane(dst)), "v"(voff), "s"(r), "s"(__builtin_amdgcn_readfirstlane(half))
        : "memory", "scc");
}
// The same load without the compiler-level memory barrier, for loads interleaved with the matrix instructions of a
// loop whose LDS reads must stay free to move (they touch a different LDS image; the s_barrier / counted wait that
// order the image's reuse are themselves compiler barriers).
__device__ __forceinline__ void dma_load16_relaxed(dma_rsrc_t r, char* lds_wave_base, unsigned voff, unsigned soff) {
    const unsigned dst = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)lds_wave_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(__builtin_amdgcn_readfirstlane(dst)), "v"(voff), "s"(r), "s"(soff));
}
// at most N of this wave's DMA loads still in flight
template <int N> __device__ __forceinline__ void dma_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0f70);
}
__device__ __forceinline__ void dma_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// this wave's LDS stores are complete (lgkmcnt(0)): whoever passes a later raw barrier sees them
__device__ __forceinline__ void lds_flush() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    asm volatile("" ::: "memory");
}
// workgroup barrier behind LDS stores of this wave only: lgkmcnt(0), DMA loads stay in flight
__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);                           // vmcnt 63, expcnt 7, lgkmcnt 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#endif

}  // namespace gpmpc
