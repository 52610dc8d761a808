// fp64 matrix-core primitives for gfx950 (CDNA4).
//
// v_mfma_f64_16x16x4_f64: one wave64 computes D[16x16] = A[16x4] B[4x16] + C.  Lane l supplies
// ONE f64 of A, A[i = l & 15][k = l >> 4], and one of B, B[k = l >> 4][j = l & 15]; it receives four
// results, register r being D[row][col = l & 15] with row = (l >> 4) + 4 r (the gfx950 f64 map,
// /opt/skills/guides/cdna_hip_programming.md section 3 -- NOT the f32 map 4 (l >> 4) + r).  The library verifies
// the map on the device at start-up (gpmpc_mfma_selftest) and passes the verified `crow_mode` to
// every kernel, so a silent layout mismatch cannot produce wrong factors.
#pragma once
#include <hip/hip_runtime.h>

typedef double d4 __attribute__((ext_vector_type(4)));

// the one dynamic LDS array of the library (chain / worker kernels, DMA-staged GEMM)
#ifdef GPMPC_EMULATED
#define GPMPC_DYN_SMEM() ((double*)::emu::dyn_smem())
#else
extern __shared__ __attribute__((aligned(16))) double gpmpc_dyn_smem[];
#define GPMPC_DYN_SMEM() (gpmpc_dyn_smem)
#endif

namespace gpmpc {

__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// D = C - A B: for the f64 matrix instruction the BLGP field carries NEG modifiers (bit 0: A), so the subtraction of a
// trailing update costs no VALU instruction (the workers negated their A fragments with two v_xor per K pair before r05)
__device__ __forceinline__ d4 mfma16_nega(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1);
}

// row inside the 16x16 result tile of accumulator register r held by `lane`
__device__ __forceinline__ int crow(int lane, int r, int mode) {
    return mode == 0 ? (lane >> 4) + 4 * r : 4 * (lane >> 4) + r;
}

// broadcast a double from a wave-uniform (compile-time after unrolling) source lane through SGPRs:
// two v_readlane_b32, no LDS crossbar traffic
__device__ __forceinline__ double bcast(double v, int src_lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

}  // namespace gpmpc
