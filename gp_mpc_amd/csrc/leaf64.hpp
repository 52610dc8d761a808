// Leaf of the recursive Cholesky / triangular-inverse: one workgroup factors a 64 x 64 diagonal
// block held in LDS and inverts the factor (a4 + a6 on a diagonal block; np.linalg.cholesky
// optimize.py:346, invL optimize.py:489).
//
// Inside the block the same right-looking scheme runs at 16-column granularity.  The sequential
// chain is one wave factoring a 64 x 16 PANEL with a row per lane (all 64 lanes busy): column j is
// scaled by rsqrt(pivot) and the rank-1 update of the remaining panel columns uses SGPR broadcasts
// (v_readlane) of the diagonal block's rows -- the sub-panel triangular solve falls out of the same
// instruction stream, with no LDS round trips or barriers inside the panel.  The rank-16 update of
// the trailing tiles and the assembly of the 64 x 64 inverse are v_mfma_f64_16x16x4_f64 products on
// LDS-resident operands; the 16 x 16 diagonal inverses are computed by an otherwise idle wave while
// the next panel is being factored.
#pragma once
#include "mfma_f64.hpp"

namespace gpmpc {

constexpr int LS = 65;  // padded LDS row stride (doubles)

// acc += Aop(16 x K) * Bop(K x 16); Aop(i,k) = Sa[(ar+i)*LS + ac + k];
// Bop(k,j) = BT ? Sb[(br+j)*LS + bc + k] : Sb[(br+k)*LS + bc + j]
template <bool BT>
__device__ __forceinline__ d4 lds_mm16(const double* Sa, int ar, int ac, const double* Sb, int br, int bc,
                                       int K, int lane, d4 acc) {
    const int fr = lane & 15, fk = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const double a = Sa[(ar + fr) * LS + ac + k0 + fk];
        const double b = BT ? Sb[(br + fr) * LS + bc + k0 + fk] : Sb[(br + k0 + fk) * LS + bc + fr];
        acc = mfma16(a, b, acc);
    }
    return acc;
}

// One product with K known at compile time: every fragment is requested before the first matrix instruction (the loop
// above is "two LDS reads, wait, one matrix instruction" per step of 4 in K).  Same summation order: same bits.
template <bool BT, int K>
__device__ __forceinline__ d4 lds_mm16k(const double* Sa, int ar, int ac, const double* Sb, int br, int bc, int lane, d4 acc) {
    const int fr = lane & 15, fk = lane >> 4;
    double a[K / 4], b[K / 4];
#pragma unroll
    for (int q = 0; q < K / 4; ++q) {
        a[q] = Sa[(ar + fr) * LS + ac + 4 * q + fk];
        b[q] = BT ? Sb[(br + fr) * LS + bc + 4 * q + fk] : Sb[(br + 4 * q + fk) * LS + bc + fr];
    }
#pragma unroll
    for (int q = 0; q < K / 4; ++q) acc = mfma16(a[q], b[q], acc);
    return acc;
}

// Two independent products of one wave, K known at compile time: all fragments are fetched first, then the matrix
// instructions of the two dependency chains alternate.  lds_mm16's loop is "two LDS reads, wait, one matrix instruction that
// depends on the previous one", ~200 cycles per step of 4 in K of which the instruction itself is 64 (in-kernel stamps, r03:
// the chain's two K = 64 diagonal-update tiles per wave took 3.0 us, 2.2 us this way; for ONE chain, and for the K = 16
// products of the leaf, fetching first measured slower).  Same order of the summation per product: bit-identical results.
// (bc0 / bc1 >= 0: the two products' B operands start at different columns, bc is ignored)
template <bool BT, int K>
__device__ __forceinline__ void lds_mm16k_x2(const double* Sa, int ar0, int ar1, int ac, const double* Sb, int br0, int br1,
                                             int bc, int lane, d4& acc0, d4& acc1, int bc0 = -1, int bc1 = -1) {
    const int fr = lane & 15, fk = lane >> 4;
    if (bc0 < 0) { bc0 = bc; bc1 = bc; }
    double a0[K / 4], b0[K / 4], a1[K / 4], b1[K / 4];
#pragma unroll
    for (int q = 0; q < K / 4; ++q) {
        a0[q] = Sa[(ar0 + fr) * LS + ac + 4 * q + fk];
        b0[q] = BT ? Sb[(br0 + fr) * LS + bc0 + 4 * q + fk] : Sb[(br0 + 4 * q + fk) * LS + bc0 + fr];
        a1[q] = Sa[(ar1 + fr) * LS + ac + 4 * q + fk];
        b1[q] = BT ? Sb[(br1 + fr) * LS + bc1 + 4 * q + fk] : Sb[(br1 + 4 * q + fk) * LS + bc1 + fr];
    }
#pragma unroll
    for (int q = 0; q < K / 4; ++q) {
        acc0 = mfma16(a0[q], b0[q], acc0);
        acc1 = mfma16(a1[q], b1[q], acc1);
    }
}

__device__ __forceinline__ void lds_put16(double* S, int r0, int c0, d4 v, double scale, int lane, int mode) {
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(r0 + crow(lane, r, mode)) * LS + c0 + (lane & 15)] = scale * v[r];
}

__device__ __forceinline__ void lds_sub16(double* S, int r0, int c0, d4 v, int lane, int mode) {
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(r0 + crow(lane, r, mode)) * LS + c0 + (lane & 15)] -= v[r];
}

// One wave: inverse of the 16 x 16 lower-triangular block at (o,o) of S into T.  Lane r (< 16) owns row r
// of the block in registers and builds COLUMN r of the inverse by forward substitution; l_ik comes from
// lane i's register a[k] through an SGPR broadcast.  Lanes >= 16 shadow rows r & 15 so that every lane
// executes the same broadcasts.  Drinv: reciprocal pivots left in LDS by panel_potrf (no divisions on the
// factorisation path); nullptr (inverse-only entry, gpmpc_set_factors) divides.
__device__ __forceinline__ void inv16(const double* S, double* T, int o, int lane, const double* Drinv) {
    const int r = lane & 15;
    double a[16], rinv[16], x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = (c <= r) ? S[(o + r) * LS + o + c] : 0.0;
    if (Drinv) {
#pragma unroll
        for (int j = 0; j < 16; ++j) rinv[j] = Drinv[o + j];
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) rinv[j] = 1.0 / bcast(a[j], j);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double s = (i == r) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= bcast(a[k], i) * x[k];
        x[i] = s * rinv[i];
    }
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) T[(o + i) * LS + o + r] = x[i];
    }
}

// One wave: Cholesky of the 64 x 16 panel = rows 16t..63 of columns 16t..16t+15 of S (LDS), lane = row.
// Returns the first non-positive pivot column (0-based within the panel) or -1.
// The panel index is a template parameter: the source lanes of the ~270 v_readlane broadcasts per panel are then
// instruction immediates instead of SGPRs computed from a loop variable (8.5 instead of 10.8 us for the four panels of
// a leaf, tools/ubench/leaf_loop_bench.hip).
template <int TT>
__device__ __forceinline__ int panel_potrf(double* S, double* Drinv, int lane) {
    constexpr int o = 16 * TT;
    const int r = lane;
    double a[16], rinv[16];
    int bad = -1;
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = S[r * LS + o + c];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double ajj = bcast(a[j], o + j);
        if (!(ajj > 0.0) && bad < 0) bad = j;  // also catches NaN; wave-uniform
        const double ri = rsqrt(ajj);
        rinv[j] = ri;
        const double lj = (r == o + j) ? ajj * ri : a[j] * ri;
        a[j] = lj;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) a[k] -= lj * bcast(lj, o + k);
    }
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (lane == c) Drinv[o + c] = rinv[c];   // 1 / L_cc for the diagonal-inverse wave
    if (r >= o) {
#pragma unroll
        for (int c = 0; c < 16; ++c) S[r * LS + o + c] = (r - o >= 16 || c <= r - o) ? a[c] : 0.0;
    }
    return bad;
}

// ---- DPP row broadcasts (gfx90a+: `row_newbcast:K` is the one DPP control the 64-bit ALU accepts) ------------------
// rowb<K>(v): lane l receives the value lane (l & ~15) + K holds -- a broadcast inside each group of 16 lanes without the
// SGPR round trip of v_readlane (two readlanes + their SGPR-write latency per double).  fnma_rowb<K>(acc, b, own):
// acc -= rowb<K>(b) * own as ONE v_fmac_f64_dpp.  The hazard "VALU writes a VGPR, a DPP operand reads it within 2 wait
// states" is invisible to the compiler inside inline assembly, so every statement starts with the s_nop that covers it.
#ifdef GPMPC_EMULATED
template <int K> __device__ __forceinline__ double rowb(double v) { return emu::wave_xchg(v, (emu::lane() & ~15) + K); }
template <int K, bool NOP = true> __device__ __forceinline__ void fnma_rowb(double& acc, double b, double own) {
    acc = __builtin_fma(-rowb<K>(b), own, acc);
}
#else
template <int K> __device__ __forceinline__ double rowb(double v) {
    return __longlong_as_double(__builtin_amdgcn_update_dpp(0ll, __double_as_longlong(v), 0x150 + K, 0xf, 0xf, true));
}
template <int K, bool NOP = true> __device__ __forceinline__ void fnma_rowb(double& acc, double b, double own) {
    if constexpr (NOP)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(b), "v"(own), "n"(K));
    else    // the caller guarantees that `b` was not written by one of the two preceding VALU instructions
        asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(b), "v"(own), "n"(K));
}
#endif

// rsqrt of a positive double: v_rsq_f64 (about 2^-26) + one third-order correction.
__device__ __forceinline__ double rsqrt_newton(double x) {
#ifdef GPMPC_EMULATED
    return 1.0 / std::sqrt(x);
#else
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-x * y0, y0, 1.0);                       // 1 - x y0^2
    return __builtin_fma(y0 * e, __builtin_fma(e, 0.375, 0.5), y0);         // y0 (1 + e/2 + 3 e^2/8)
#endif
}

// One wave: Cholesky of the 64 x 16 panel, DPP form.  Lane l owns panel row l in a[] (as panel_potrf) AND row o + (l & 15)
// of the 16 x 16 diagonal block in d[] -- the diagonal block is replicated in the four 16-lane groups so that
// row_newbcast:k delivers L_kj to every lane.  Column j: x = pivot (rowb<j>), y = rsqrt(x), d[j] *= y, a[j] *= y, then for
// k > j: d[k] -= L_kj d[j], a[k] -= L_kj a[j] as two v_fmac_f64_dpp (MODE 2; MODE 1 = v_mov_b64_dpp + two v_fma_f64, the
// compiler-scheduled form kept for the micro-benchmark).  Measured (tools/ubench/panel_dpp_bench.hip, one wave, idle
// chip): a v_fmac_f64_dpp issues like a plain v_fma_f64 (2.8 ns), "two v_readlane + fma" costs 9.8 ns.
// No per-column test of the pivot: a non-positive or NaN pivot makes y NaN (rsq of a negative; 0 x inf in the
// correction), the NaN spreads to every later column, so ONE test of the last y decides, and only then the stored
// reciprocals are scanned for the first bad column (same `info` as the per-column test of panel_potrf).
// o = 16 x panel index at RUN time: the leaf then holds this code once, not four times (the chain kernel's 55 KB of
// straight-line code do not stay in the 64 KB instruction cache it shares with a neighbour CU, and refetching them through
// a busy L2 cost the leaf ~3 us per call: tools/ubench/leaf_icache_bench.hip).
// T (r05; o >= 16 only): the INVERSE of the panel's 16 x 16 diagonal block for free.  What the panel does to a lane's row,
// a <- a L_oo^-T (column j scaled by 1 / L_jj, then a[k] -= L_kj a[j]), is a forward substitution; lanes 0-15 hold rows
// ABOVE the panel when o >= 16 -- zeros of S's upper triangle, dead weight in every instruction -- so they start from the
// unit vectors e_i instead and end with e_i L_oo^-T = column i of L_oo^-1, which goes to T[o + c][o + i].  Not one extra
// VALU instruction; the unit entries pass through S's (otherwise zero, never read) block (0, o / 16) for the loads.  The
// last block's inverse used to take an idle wave 1.31 us AFTER the last panel (profiles/r04_chain_trace.txt).
template <int MODE>
__device__ __forceinline__ int panel_potrf_dpp_at(double* S, double* Drinv, int lane, const int o, double* T = nullptr) {
    const int r = lane, i = lane & 15;
    double a[16], d[16];
    const bool unit_rows = T != nullptr && o >= 16;            // (wave-uniform)
    // PRECONDITION of the free inverse: S's block (0, o / 16) -- rows 0-15, columns o .. o + 15 -- holds exact zeros (strictly
    // upper part of the 64 x 64 tile).  Every caller stages the LOWER triangle and zeroes the rest (leaf64_kernel, the chain's
    // initial / non-prefetched loads; the prefetched path never writes above the diagonal); a caller that staged the full
    // symmetric tile would get a wrong inv_kk silently.  The emulated build checks it.
#ifdef GPMPC_EMULATED
    if (unit_rows && lane < 16)
        for (int c = 0; c < 16; ++c)
            if (S[lane * LS + o + c] != 0.0) { fprintf(stderr, "leaf64: S above the diagonal is not zero (row %d, column %d)\n", lane, o + c); abort(); }
#endif
    if (unit_rows && lane < 16) S[lane * LS + o + lane] = 1.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) { a[c] = S[r * LS + o + c]; d[c] = S[(o + i) * LS + o + c]; }
    if (unit_rows && lane < 16) S[lane * LS + o + lane] = 0.0;  // (the same wave's LDS operations complete in order)
    double y = 0.0;
#define GPMPC_PANEL_COL(j)                                                                              \
    {                                                                                                     \
        const double x = rowb<j>(d[j]);                                                                   \
        y = rsqrt_newton(x);                                                                              \
        Drinv[o + j] = y;      /* uniform value, uniform address: 1 / L_jj for the diagonal-inverse wave */ \
        d[j] *= y;                                                                                        \
        a[j] *= y;                                                                                        \
        GPMPC_PANEL_UPD(j, 1) GPMPC_PANEL_UPD(j, 2) GPMPC_PANEL_UPD(j, 3) GPMPC_PANEL_UPD(j, 4) GPMPC_PANEL_UPD(j, 5)   \
        GPMPC_PANEL_UPD(j, 6) GPMPC_PANEL_UPD(j, 7) GPMPC_PANEL_UPD(j, 8) GPMPC_PANEL_UPD(j, 9) GPMPC_PANEL_UPD(j, 10)  \
        GPMPC_PANEL_UPD(j, 11) GPMPC_PANEL_UPD(j, 12) GPMPC_PANEL_UPD(j, 13) GPMPC_PANEL_UPD(j, 14) GPMPC_PANEL_UPD(j, 15) \
    }
#define GPMPC_PANEL_UPD(j, k)                                                                            \
    if constexpr (k > j) {                                                                                \
        if constexpr (MODE == 2) {                                                                        \
            fnma_rowb<k, k == j + 1>(d[k], d[j], d[j]);   /* (asm volatile keeps the statement order) */   \
            fnma_rowb<k, false>(a[k], d[j], a[j]);                                                        \
        } else {                                                                                          \
            const double b = rowb<k>(d[j]);                                                               \
            d[k] = __builtin_fma(-b, d[j], d[k]);                                                         \
            a[k] = __builtin_fma(-b, a[j], a[k]);                                                         \
        }                                                                                                 \
    }
    GPMPC_PANEL_COL(0) GPMPC_PANEL_COL(1) GPMPC_PANEL_COL(2) GPMPC_PANEL_COL(3) GPMPC_PANEL_COL(4) GPMPC_PANEL_COL(5)
    GPMPC_PANEL_COL(6) GPMPC_PANEL_COL(7) GPMPC_PANEL_COL(8) GPMPC_PANEL_COL(9) GPMPC_PANEL_COL(10) GPMPC_PANEL_COL(11)
    GPMPC_PANEL_COL(12) GPMPC_PANEL_COL(13) GPMPC_PANEL_COL(14) GPMPC_PANEL_COL(15)
#undef GPMPC_PANEL_COL
#undef GPMPC_PANEL_UPD
    if (r >= o) {
#pragma unroll
        for (int c = 0; c < 16; ++c) S[r * LS + o + c] = (r - o >= 16 || c <= r - o) ? a[c] : 0.0;
    } else if (unit_rows && lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) T[(o + c) * LS + o + lane] = a[c];     // column `lane` of the block's inverse
    }
    int bad = -1;
    if (!(y * 0.0 == 0.0)) {                 // NaN or infinity in the last reciprocal (wave-uniform, rare)
        for (int c = 15; c >= 0; --c) {
            const double yc = Drinv[o + c];
            if (!(yc * 0.0 == 0.0)) bad = c;
        }
    }
    return bad;
}

template <int TT, int MODE>
__device__ __forceinline__ int panel_potrf_dpp(double* S, double* Drinv, int lane) {
    return panel_potrf_dpp_at<MODE>(S, Drinv, lane, 16 * TT);
}

// One wave: inverse of the 16 x 16 lower-triangular block at (o,o) of S into T, DPP form of inv16: lane (q, r) owns row r of
// the block (replicated in the four 16-lane groups) and builds column r of the inverse by forward substitution,
// x_i = (delta_ir - sum_{k<i} L_ik x_k) / L_ii with L_ik = rowb<i>(a[k]) folded into one v_fmac_f64_dpp.
__device__ __forceinline__ void inv16_dpp(const double* S, double* T, int o, int lane, const double* Drinv) {
    const int r = lane & 15;
    double a[16], x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = (c <= r) ? S[(o + r) * LS + o + c] : 0.0;
#define GPMPC_INV_TERM(i, k) if constexpr (k < i) fnma_rowb<i, false>(s, a[k], x[k]);
#define GPMPC_INV_ROW(i)                                                                                 \
    {                                                                                                     \
        double s = (i == r) ? 1.0 : 0.0;                                                                  \
        GPMPC_INV_TERM(i, 0) GPMPC_INV_TERM(i, 1) GPMPC_INV_TERM(i, 2) GPMPC_INV_TERM(i, 3) GPMPC_INV_TERM(i, 4)          \
        GPMPC_INV_TERM(i, 5) GPMPC_INV_TERM(i, 6) GPMPC_INV_TERM(i, 7) GPMPC_INV_TERM(i, 8) GPMPC_INV_TERM(i, 9)          \
        GPMPC_INV_TERM(i, 10) GPMPC_INV_TERM(i, 11) GPMPC_INV_TERM(i, 12) GPMPC_INV_TERM(i, 13) GPMPC_INV_TERM(i, 14)     \
        x[i] = s * Drinv[o + i];                                                                          \
    }
    // (the DPP operands a[k] are loaded once, long before their first use: no hazard inside the substitution)
    GPMPC_INV_ROW(0) GPMPC_INV_ROW(1) GPMPC_INV_ROW(2) GPMPC_INV_ROW(3) GPMPC_INV_ROW(4) GPMPC_INV_ROW(5) GPMPC_INV_ROW(6)
    GPMPC_INV_ROW(7) GPMPC_INV_ROW(8) GPMPC_INV_ROW(9) GPMPC_INV_ROW(10) GPMPC_INV_ROW(11) GPMPC_INV_ROW(12)
    GPMPC_INV_ROW(13) GPMPC_INV_ROW(14) GPMPC_INV_ROW(15)
#undef GPMPC_INV_ROW
#undef GPMPC_INV_TERM
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) T[(o + i) * LS + o + r] = x[i];
    }
}

// The factor + invert body shared by leaf64_kernel and the persistent chain kernel (chol_chain.hpp).
// 256 threads; S holds the (lower) 64 x 64 block on entry and L on exit, T receives L^-1 (its strictly
// upper 16 x 16 blocks must be zero on entry and stay zero), U and Dr are scratch.  Returns the first
// non-positive pivot column (0-based) or -1 (meaningful in wave 0).
// `phases` (bit 0 panel, 1 rank-16 update, 2 diagonal inverses, 3 inverse assembly) exists for the
// micro-benchmark tools/ubench/leaf_bench.hip only; the library always passes 15.
//
// `hook` lets the caller slip work of its own behind the factorisation: hook.before() runs (all threads)
// just before the barrier that follows the third 16-column panel, hook.after() right after it -- about
// two thirds into the leaf -- and hook.land() in waves 2 and 3 while wave 0 factors the last panel.  The chain kernel
// uses them to poll a flag (before), to issue the global loads of its next tiles in waves 2 and 3 (after) and to put
// what they fetched into LDS (land): the latency hides behind the last panel and no register is held across the leaf.
// hook.first() runs in waves 1-3 while wave 0 factors the FIRST panel (columns 0-15 of S): the chain kernel finishes the
// trailing update of the block's columns 16-63 there (only the first 16 columns are needed before the leaf starts).
#ifndef GPMPC_LEAF_DPP
#define GPMPC_LEAF_DPP 1
#endif
#if GPMPC_LEAF_DPP
#define GPMPC_INV16 inv16_dpp
#else
#define GPMPC_INV16 inv16
#endif
struct LeafNoHook {
    __device__ __forceinline__ void first() {}
    __device__ __forceinline__ void before() {}
    __device__ __forceinline__ void after() {}
    __device__ __forceinline__ void land() {}
    __device__ __forceinline__ void stamp(int) {}
};

template <class Hook>
__device__ __forceinline__ int leaf_body(double* S, double* T, double* U, double* Dr, int do_chol, int phases,
                                         int crow_mode, Hook& hook) {
    const int tid = threadIdx.x, lane = tid & 63;
    // (wave-uniform by construction; said so to the compiler: the per-wave roles below become scalar branches, and what
    //  one role keeps in registers -- the chain kernel's prefetch in waves 2 and 3 -- is not live in the others' code)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bad = -1;
    hook.stamp(0);
#if GPMPC_LEAF_DPP
    if (do_chol) {
        // r05 schedule.  Wave 0 factors the four 16-column panels; panels 1-3 leave the inverse of their diagonal block in
        // T at no cost (panel_potrf_dpp_at: unit rows in the lanes above the panel), block 0's is wave 1's job next to
        // panel 1 as before.  The assembly of the 64 x 64 inverse
        //     [[A, 0], [C, B]]^-1 = [[A^-1, 0], [-B^-1 (C A^-1), B^-1]]        (32-blocks from 16-blocks, 64 from 32)
        // no longer waits for the last panel: every product whose operands are final runs on the waves that idle next to
        // wave 0's panels (the S columns of panel t are final behind panel t's barrier):
        //   next to panel 1 (wave 1):  inv00
        //   next to panel 2 (wave 1):  U1 = L10 inv00,  T10 = -inv11 U1,  W[:, 0:16] = L[32:64, 0:32] [inv00; T10]      (wave 2: W[:, 16:32] = L[32:64, 16:32] inv11)
        //   next to panel 3 (wave 1):  U2 = L32 inv22,   T[32:48, 0:32] = -inv22 W[0:16, :]
        // and behind the last panel only  T32 = -inv33 U2  (wave 0) next to  acc = inv33 W[16:32, :]  (waves 1, 2), a barrier,
        // T[48:64, 0:32] = -(acc + T32 W[0:16, :]): two short products on the critical path instead of the last block's
        // substitution + four (r04: 1.31 + 0.67 + 0.91 us per leaf, profiles/r04_chain_trace.txt).  W = C A^-1 of the 64-level
        // lives in U[32:64, 0:32], U1 in U[16:32, 0:16], U2 in U[48:64, 32:48]; U is free from the barrier behind panel 0 on
        // (the chain kernel's hook.first() reads it next to panel 0).
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            if (wave == 0) {
                if (phases & 1) {
                    const int b = panel_potrf_dpp_at<2>(S, Dr, lane, 16 * t, T);
                    if (b >= 0 && bad < 0) bad = 16 * t + b;
                }
            } else if (t == 0) {
                hook.first();                           // columns 16-63 of S may still be written here (panel 0 owns 0-15)
            } else if (wave == 1 && (phases & 8)) {
                d4 z = d4{0.0, 0.0, 0.0, 0.0};
                if (t == 1) {
                    inv16_dpp(S, T, 0, lane, Dr);
                } else if (t == 2) {
                    lds_put16(U, 16, 0, lds_mm16<false>(S, 16, 0, T, 0, 0, 16, lane, z), 1.0, lane, crow_mode);
                    lds_put16(T, 16, 0, lds_mm16<false>(T, 16, 16, U, 16, 0, 16, lane, z), -1.0, lane, crow_mode);
                    d4 w0 = z, w1 = z;
                    lds_mm16k_x2<false, 32>(S, 32, 48, 0, T, 0, 0, 0, lane, w0, w1);      // rows 32-47 / 48-63 x [inv00; T10]
                    lds_put16(U, 32, 0, w0, 1.0, lane, crow_mode);
                    lds_put16(U, 48, 0, w1, 1.0, lane, crow_mode);
                } else {
                    lds_put16(U, 48, 32, lds_mm16<false>(S, 48, 32, T, 32, 32, 16, lane, z), 1.0, lane, crow_mode);
                    d4 f0 = z, f1 = z;
                    lds_mm16k_x2<false, 16>(T, 32, 32, 32, U, 32, 32, 0, lane, f0, f1, 0, 16);   // inv22 x W[0:16, 0:16 | 16:32]
                    lds_put16(T, 32, 0, f0, -1.0, lane, crow_mode);
                    lds_put16(T, 32, 16, f1, -1.0, lane, crow_mode);
                }
            } else if (wave == 2 && t == 2 && (phases & 8)) {
                d4 w0 = d4{0.0, 0.0, 0.0, 0.0}, w1 = w0;
                lds_mm16k_x2<false, 16>(S, 32, 48, 16, T, 16, 16, 16, lane, w0, w1);      // rows 32-47 / 48-63 of L[:, 16:32] x inv11
                lds_put16(U, 32, 16, w0, 1.0, lane, crow_mode);
                lds_put16(U, 48, 16, w1, 1.0, lane, crow_mode);
            } else if (wave >= 2 && t == 3) {
                hook.land();                            // waves 2 and 3 have nothing else to do behind the last panel
            }
            // the part of panel t-1's rank-16 update that panel t does not need (tiles (i, j), j > t: see below), on the waves
            // without a panel: wave 3 (and wave 2 next to panel 1, where it has nothing else)
            if (t >= 1 && t <= 2 && wave >= 2 && (phases & 2)) {
                const int op = 16 * (t - 1);
                int cnt = 0;
                for (int i = t + 1; i <= 3; ++i)
                    for (int j = t + 1; j <= i; ++j, ++cnt)
                        if ((t == 1 ? 2 + (cnt & 1) : 3) == wave) {
                            d4 pacc = d4{0.0, 0.0, 0.0, 0.0};
                            pacc = lds_mm16<true>(S, 16 * i, op, S, 16 * j, op, 16, lane, pacc);
                            lds_sub16(S, 16 * i, 16 * j, pacc, lane, crow_mode);
                        }
            }
            hook.stamp(1 + 3 * t);
            if (t == 2) hook.before();
            __syncthreads();
            hook.stamp(2 + 3 * t);
            if (t == 2) hook.after();
            // rank-16 update A_ij -= L_it L_jt^T, t < j <= i <= 3, in two parts (r05): only block column t + 1 -- what the next
            // panel factors -- here, one tile per wave, on the leaf's critical path; the tiles (i, j), j > t + 1, next to that
            // panel (above).  Every tile still receives its updates in the order of the panels: same bits as one pass.
            const int o = 16 * t;
            if (t < 3 && (phases & 2) && wave <= 2 - t) {
                const int i = t + 1 + wave, j = t + 1;
                d4 pacc = d4{0.0, 0.0, 0.0, 0.0};
                pacc = lds_mm16<true>(S, 16 * i, o, S, 16 * j, o, 16, lane, pacc);
                lds_sub16(S, 16 * i, 16 * j, pacc, lane, crow_mode);
            }
            if (t < 3) __syncthreads();
            if (t < 3) hook.stamp(3 + 3 * t);
        }
        hook.stamp(12);
        if (phases & 8) {
            d4 acc = d4{0.0, 0.0, 0.0, 0.0};
            if (wave == 0) lds_put16(T, 48, 32, lds_mm16<false>(T, 48, 48, U, 48, 32, 16, lane, acc), -1.0, lane, crow_mode);
            else if (wave <= 2) acc = lds_mm16<false>(T, 48, 48, U, 48, 16 * (wave - 1), 16, lane, acc);
            __syncthreads();
            hook.stamp(13);
            if (wave == 1 || wave == 2) {
                acc = lds_mm16<false>(T, 48, 32, U, 32, 16 * (wave - 1), 16, lane, acc);
                lds_put16(T, 48, 16 * (wave - 1), acc, -1.0, lane, crow_mode);
            }
            __syncthreads();
            hook.stamp(14);
        }
        return bad;
    }
#endif
    if (do_chol) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (wave == 0 && (phases & 1)) {
                const int b = t == 0 ? panel_potrf<0>(S, Dr, lane) : t == 1 ? panel_potrf<1>(S, Dr, lane)
                            : t == 2 ? panel_potrf<2>(S, Dr, lane) : panel_potrf<3>(S, Dr, lane);
                if (b >= 0 && bad < 0) bad = 16 * t + b;
            } else if (wave == 1 && t >= 1 && (phases & 4)) {
                GPMPC_INV16(S, T, 16 * (t - 1), lane, Dr);   // inverse of the previous diagonal block
            } else if (wave >= 2 && t == 3) {
                hook.land();
            } else if (wave >= 1 && t == 0) {
                hook.first();
            }
            hook.stamp(1 + 3 * t);
            if (t == 2) hook.before();
            __syncthreads();
            hook.stamp(2 + 3 * t);
            if (t == 2) hook.after();
            const int o = 16 * t;
            int cnt = 0;
            for (int i = t + 1; i <= 3 && (phases & 2); ++i)
                for (int j = t + 1; j <= i; ++j, ++cnt)
                    if ((cnt & 3) == wave) {
                        d4 pacc = d4{0.0, 0.0, 0.0, 0.0};
                        pacc = lds_mm16<true>(S, 16 * i, o, S, 16 * j, o, 16, lane, pacc);
                        lds_sub16(S, 16 * i, 16 * j, pacc, lane, crow_mode);
                    }
            if (t < 3) __syncthreads();
            if (t < 3) hook.stamp(3 + 3 * t);
        }
        if (wave == 1 && (phases & 4)) GPMPC_INV16(S, T, 48, lane, Dr);
        __syncthreads();
        hook.stamp(12);
    } else {
        if (wave < 4) inv16(S, T, 16 * wave, lane, nullptr);
        __syncthreads();
    }
    // assemble the 64 x 64 inverse from the four 16 x 16 diagonal inverses:
    // inv21 = -inv22 (L21 inv11), first for the two 32-blocks, then for the 64-block
    if (phases & 8) {
        const int c1 = 32 * wave, r2 = c1 + 16;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
        if (wave < 2) acc = lds_mm16<false>(S, r2, c1, T, c1, c1, 16, lane, acc);
        if (wave < 2) lds_put16(U, r2, c1, acc, 1.0, lane, crow_mode);
        __syncthreads();
        acc = d4{0.0, 0.0, 0.0, 0.0};
        if (wave < 2) acc = lds_mm16<false>(T, r2, r2, U, r2, c1, 16, lane, acc);
        if (wave < 2) lds_put16(T, r2, c1, acc, -1.0, lane, crow_mode);
        __syncthreads();
        hook.stamp(13);
    }
    if (phases & 8) {
        const int pi = (wave >> 1) & 1, pj = wave & 1;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
        acc = lds_mm16<false>(S, 32 + 16 * pi, 0, T, 0, 16 * pj, 32, lane, acc);
        lds_put16(U, 32 + 16 * pi, 16 * pj, acc, 1.0, lane, crow_mode);
        __syncthreads();
        acc = d4{0.0, 0.0, 0.0, 0.0};
        acc = lds_mm16<false>(T, 32 + 16 * pi, 32, U, 32, 16 * pj, 32, lane, acc);
        __syncthreads();
        lds_put16(T, 32 + 16 * pi, 16 * pj, acc, -1.0, lane, crow_mode);
        __syncthreads();
        hook.stamp(14);
    }
    return bad;
}

__device__ __forceinline__ int leaf_body(double* S, double* T, double* U, double* Dr, int do_chol, int phases,
                                         int crow_mode) {
    LeafNoHook h;
    return leaf_body(S, T, U, Dr, do_chol, phases, crow_mode, h);
}

// grid (nblk, 1, batch), 256 threads: workgroup x handles the diagonal block starting at row/column
// off + 64 x (nblk > 1 only for the inverse-only mode, where the blocks are independent).
// Ain: source of the diagonal block (the running K for a factorisation, L itself for inverse-only);
// L / Inv: destinations.  All are [batch][ld x ld] row-major.
__global__ void __launch_bounds__(256) leaf64_kernel(const double* Ain, double* L, double* Inv, long ld,
                                                     long sBatch, int off0, int do_chol, int* info,
                                                     int crow_mode, int phases = 15) {
    const int off = off0 + 64 * (int)blockIdx.x;
    __shared__ double S[64 * LS];
    __shared__ double T[64 * LS];
    __shared__ double U[64 * LS];
    __shared__ double Dr[64];
    const int tid = threadIdx.x;
    const long base = (long)blockIdx.z * sBatch + (long)off * ld + off;
    const double* __restrict__ src = Ain + base;
    for (int idx = tid; idx < 4096; idx += 256) {
        const int rr = idx >> 6, cc = idx & 63;
        S[rr * LS + cc] = (cc <= rr) ? src[(long)rr * ld + cc] : 0.0;
        T[rr * LS + cc] = 0.0;
    }
    __syncthreads();
    const int bad = leaf_body(S, T, U, Dr, do_chol, phases, crow_mode);
    if (do_chol && tid == 0 && bad >= 0) atomicCAS(&info[blockIdx.z], 0, off + bad + 1);
    double* __restrict__ dl = L + base;
    double* __restrict__ di = Inv + base;
    for (int idx = tid; idx < 4096; idx += 256) {
        const int rr = idx >> 6, cc = idx & 63;
        if (do_chol) dl[(long)rr * ld + cc] = (cc <= rr) ? S[rr * LS + cc] : 0.0;
        di[(long)rr * ld + cc] = (cc <= rr) ? T[rr * LS + cc] : 0.0;
    }
}

}  // namespace gpmpc
