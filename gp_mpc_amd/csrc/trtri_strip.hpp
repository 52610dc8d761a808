// Inverse of the lower-triangular diagonal blocks of L by COLUMN STRIPS: one launch instead of the level-by-level doubling.
//
// a6 (optimize.py:489-490: invL = solve(L, I)) is formed panel by panel; what every panel step needs first is the inverse I_i
// of its own diagonal block T = L[P_i, P_i] (512 ... 2048 rows).  Rounds 1-5 built it by doubling -- for s = 64, 128, ...:
// W = T21 inv11, inv21 = -inv22 W as two batched GEMM launches per level -- i.e. 2 log2(n / 64) DEPENDENT launches of tiny
// products (n = 1024: eight launches, 0.13 GFLOP in the largest, 180-240 us on MI355X next to the factorisation: launch
// gaps and 32-step K loops in 32 x 32 tiles; profiles/r06_two_calls_step_timeline.txt).  Behind the chain kernel those
// launches ARE the tail of the fit (I_2, then one product), and the same chain of launches delays the row-panel products
// of the panels before.
//
// Columns of a triangular inverse do not depend on each other: X = T^-1 solves T X = I column by column,
//     X[i, :] = -inv_ii sum_{m = j}^{i-1} T[i, m] X[m, :]      (64-row blocks i > j; X[j, :] = inv_jj's columns),
// with the 64 x 64 diagonal inverses inv_ii the leaf already left in Inv.  One workgroup owns a strip of 16 columns and walks
// down its block rows; its part of X lives in LDS (<= 1024 rows x 16 doubles = 128 KB) in the B-fragment order of the f64
// matrix instruction, T streams in from L2 as A fragments (the K index permuted so that a lane reads 32 contiguous bytes
// and a wave whole 128-byte lines).  n / 16 workgroups, no hand-offs between them; the longest strip issues
// n^2 / 2 x 16 x 2 / 2048 / 4 = 2048 matrix instructions per wave at n = 1024 (+ the products with inv_ii): ~60 us.
// Blocks larger than TRTRI_STRIP_N rows: strips inside every aligned TRTRI_STRIP_N-row chunk, the levels above by doubling.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include "mfma_f64.hpp"

namespace gpmpc {

constexpr int TRTRI_STRIP_W = 16;          // columns per workgroup (the N of the matrix instruction)
constexpr int TRTRI_STRIP_N = 1024;        // rows of a chunk (LDS: 16 blocks x 8 KB + two 8 KB blocks of partial sums)
constexpr int TRTRI_STRIP_THREADS = 512;   // 8 waves: row tile w = wave & 3, half of the block columns each (wave >> 2)

// LDS image of a 64 x 16 block (k x columns) as B fragments: slab sl = 16 consecutive k; inside a slab the double of
// (k = 16 sl + 4 g + kk, column c) sits at sl * 256 + (g * 16 + c) * 4 + kk, so that lane (g, c) = 16 g + c reads the four
// k of its slab as 32 contiguous bytes.
__device__ __forceinline__ int strip_idx(int k, int c) { return (k >> 4) * 256 + ((((k >> 2) & 3) * 16 + c) << 2) + (k & 3); }

struct StripFrag { double2 v[8]; };         // a lane's A fragments of one 64-column block: 4 slabs x 4 k
__device__ __forceinline__ void strip_load(StripFrag& f, const double* __restrict__ p) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        f.v[2 * sl] = *reinterpret_cast<const double2*>(p + 16 * sl);
        f.v[2 * sl + 1] = *reinterpret_cast<const double2*>(p + 16 * sl + 2);
    }
}

// grid (n / 16, batch), 512 threads, dynamic LDS (blocks of the chunk + 2) x 8 KB.  Rows / columns [base, base + n) of every matrix.
// First measured form (r06, 256 threads, one block of fragments in flight): 163 / 241 / 180 us for the three panels of C2 -- one
// wave per SIMD waiting for its L2 loads.  Now two waves per SIMD (the block columns m of a step alternate between them, their
// partial sums meet in LDS), two blocks of fragments in flight per wave, inv_ii's fragments requested at the top of the step.
__global__ void __launch_bounds__(TRTRI_STRIP_THREADS) trtri_strip_kernel(const double* __restrict__ L, double* __restrict__ Inv, long ld,
                                                                          long sM, int base, int n, int crow_mode) {
    double* Xs = GPMPC_DYN_SMEM();                               // [blocks of the strip][1024 doubles], then two blocks of partial sums
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), w = wv & 3, part = wv >> 2;
    const int c0g = TRTRI_STRIP_W * (int)blockIdx.x;             // first column of the strip inside [0, n)
    const int cb = c0g / TRTRI_STRIP_N * TRTRI_STRIP_N;          // its chunk
    const int cn = min(TRTRI_STRIP_N, n - cb), nbk = cn / 64;
    const int c0 = c0g - cb, j = c0 / 64, q = c0 % 64;
    const long off = (long)blockIdx.y * sM + (long)(base + cb) * ld + (base + cb);
    const double* __restrict__ T = L + off;
    double* __restrict__ X = Inv + off;
    double* scratch = Xs + (nbk - j) * 1024;                     // [2][1024]
    // X[j] = columns q .. q + 15 of inv_jj (already in Inv: nothing to store)
    for (int e = tid; e < 64 * TRTRI_STRIP_W; e += TRTRI_STRIP_THREADS) {
        const int r = e >> 4, c = e & 15;
        Xs[strip_idx(r, c)] = X[(long)(64 * j + r) * ld + 64 * j + q + c];
    }
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    // Every global load of a step is requested one step ahead (first measured forms: 5-9 us of exposed load latency per step
    // next to the factorisation's traffic, 15 steps): inv_ii's fragments and the step's first block of T.
    const double* __restrict__ trow = T + (long)(16 * w + fr) * ld + 4 * fg;          // + 64 i rows, + 64 m columns
    const double* __restrict__ xrow = X + (long)(16 * w + fr) * ld + 4 * fg;
    StripFrag fi, g0;                                            // of the NEXT step: inv_ii (part 0 uses it), block m = j + part
    if (j + 1 < nbk) {
        if (part == 0) strip_load(fi, xrow + (long)64 * (j + 1) * ld + 64 * (j + 1));
        if (j + part < j + 1) strip_load(g0, trow + (long)64 * (j + 1) * ld + 64 * (j + part));
    }
    for (int i = j + 1; i < nbk; ++i) {
        // rows 16 w .. 16 w + 15 of block row i:  this wave's half of  sum_m T[i, m] X[m]  (m = j + part, j + part + 2, ...)
        const double* __restrict__ arow = trow + (long)64 * i * ld;
        StripFrag fic = fi;
        d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
        StripFrag f0 = g0, f1;
        int m = j + part;
        if (m + 2 < i) strip_load(f1, arow + 64 * (m + 2));
        if (i + 1 < nbk) {                                       // next step's first requests, under this step's products
            if (part == 0) strip_load(fi, xrow + (long)64 * (i + 1) * ld + 64 * (i + 1));
            strip_load(g0, arow + (long)64 * ld + 64 * (j + part));   // (m = j + part < i + 1 always holds for part <= 1, i >= j + 1)
        }
        for (; m < i; m += 2) {
            StripFrag fc = f0;
            f0 = f1;
            if (m + 4 < i) strip_load(f1, arow + 64 * (m + 4));  // two blocks ahead of the products
            const double* __restrict__ xb = Xs + (m - j) * 1024 + lane * 4;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const double2 b0 = *reinterpret_cast<const double2*>(xb + sl * 256);
                const double2 b1 = *reinterpret_cast<const double2*>(xb + sl * 256 + 2);
                if (sl & 1) {
                    acc1 = mfma16(fc.v[2 * sl].x, b0.x, acc1);
                    acc1 = mfma16(fc.v[2 * sl].y, b0.y, acc1);
                    acc1 = mfma16(fc.v[2 * sl + 1].x, b1.x, acc1);
                    acc1 = mfma16(fc.v[2 * sl + 1].y, b1.y, acc1);
                } else {
                    acc0 = mfma16(fc.v[2 * sl].x, b0.x, acc0);
                    acc0 = mfma16(fc.v[2 * sl].y, b0.y, acc0);
                    acc0 = mfma16(fc.v[2 * sl + 1].x, b1.x, acc0);
                    acc0 = mfma16(fc.v[2 * sl + 1].y, b1.y, acc0);
                }
            }
        }
        // the two 64 x 16 partial sums become (added up by the reader) the B operand of the product with inv_ii
#pragma unroll
        for (int r = 0; r < 4; ++r) scratch[part * 1024 + strip_idx(16 * w + crow(lane, r, crow_mode), fr)] = acc0[r] + acc1[r];
        __syncthreads();
        if (part == 0) {
            // X[i] = -inv_ii acc: rows 16 w .. of the lower-triangular inv_ii reach columns < 16 (w + 1) only
            d4 xa = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                if (sl > w) break;
                const double* sp = scratch + sl * 256 + lane * 4;
                const double2 p0 = *reinterpret_cast<const double2*>(sp), p1 = *reinterpret_cast<const double2*>(sp + 2);
                const double2 q0 = *reinterpret_cast<const double2*>(sp + 1024), q1 = *reinterpret_cast<const double2*>(sp + 1026);
                xa = mfma16_nega(fic.v[2 * sl].x, p0.x + q0.x, xa);
                xa = mfma16_nega(fic.v[2 * sl].y, p0.y + q0.y, xa);
                xa = mfma16_nega(fic.v[2 * sl + 1].x, p1.x + q1.x, xa);
                xa = mfma16_nega(fic.v[2 * sl + 1].y, p1.y + q1.y, xa);
            }
            double* xi = Xs + (i - j) * 1024;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * w + crow(lane, r, crow_mode);
                xi[strip_idx(rr, fr)] = xa[r];
                X[(long)(64 * i + rr) * ld + 64 * j + q + fr] = xa[r];
            }
        }
        __syncthreads();                                         // X[i] is complete in LDS, the partial-sum blocks are free again
    }
}

inline void launch_trtri_strip(hipStream_t st, const double* L, double* Inv, long ld, long sM, int base, int n, int batch, int crow_mode) {
    constexpr int lds_max = (TRTRI_STRIP_N / 64 + 2) * 64 * TRTRI_STRIP_W * 8;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trtri_strip_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        attr_set = true;
    }
    const int lds = (std::min(n, TRTRI_STRIP_N) / 64 + 2) * 64 * TRTRI_STRIP_W * 8;       // (short strips: several workgroups per CU)
    hipLaunchKernelGGL(trtri_strip_kernel, dim3(n / TRTRI_STRIP_W, batch), dim3(TRTRI_STRIP_THREADS), lds, st, L, Inv, ld, sM, base, n, crow_mode);
}

}  // namespace gpmpc
