// Batched fp64 GEMM on the MFMA pipe with triangular-aware K ranges and fused epilogues.
//
//   C[b] = alpha * A[b] * B[b] + beta * C[b]          (epi == EPI_STORE)
//   part[b][tile_m][n] = sum_{m in tile_m} (A[b] B[b])[m][n]^2      (epi == EPI_COLSUMSQ)
//
// This one kernel is every dense contraction of the GP hot path:
//   * Cholesky panel  L21 = A21 invL11^T and trailing update A22 -= L21 L21^T   (a4, optimize.py:346)
//   * triangular inverse  invL21 = -invL22 (L21 invL11)                          (a6, optimize.py:489)
//   * K^-1 = invL^T invL                                                         (a6, optimize.py:490)
//   * w = invL y, alpha = invL^T w as N = 1 products                             (a5, optimize.py:353-354)
//   * predictive variance  sum_i (invL Ks)_ij^2                                  (a9, gp_functions.py:122-126)
//
// Design (MI355X): 256 threads = 4 waves in a 2 x 2 grid, block tile BM x BN (128, 64 or 32), K step BK.
// Operand tiles are staged global -> registers -> LDS with a one-tile software pipeline (the global
// loads of tile t+1 are in flight while tile t feeds the matrix pipe; one barrier per K step).  LDS
// holds each operand in MFMA-fragment order [k/4][row][k%4], so the 64 lanes of a fragment read
// fetch 64 consecutive doubles (512 contiguous bytes: conflict-free ds_read_b64).  With
// v_mfma_f64_16x16x4_f64 at 64 cycles/instruction the kernel is MFMA-issue bound: per K step a wave
// issues 64 (BM=128) MFMAs = 4096 cycles against 8 ds_read_b64 per 16 MFMAs and 32 KB of global
// traffic per block.
#pragma once
#include "mfma_f64.hpp"

namespace gpmpc {

enum { KA_LE_M = 1,   // A(m,k) == 0 for k > m   (A lower triangular)
       KA_GE_M = 2,   // A(m,k) == 0 for k < m   (A = T^T, T lower triangular)
       KB_LE_N = 4,   // B(k,n) == 0 for k > n   (B = T^T, T lower triangular)
       KB_GE_N = 8 }; // B(k,n) == 0 for k < n   (B lower triangular)
enum { EPI_STORE = 0, EPI_COLSUMSQ = 1 };

struct GemmP {
    const double* A;
    const double* B;
    double* C;
    long lda, ldb, ldc;
    long sA, sB, sC;      // batch strides in elements
    int M, N, K;
    double alpha, beta;
    int a_mc;             // 0: A(m,k) = A[m*lda + k] (K contiguous)   1: A(m,k) = A[k*lda + m]
    int b_nc;             // 0: B(k,n) = B[n*ldb + k] (K contiguous)   1: B(k,n) = B[k*ldb + n]
    int kflags;
    int lower;            // 1: write only n <= m (tiles strictly above the diagonal are skipped)
    int epi;
    double* part;         // EPI_COLSUMSQ: part[b*sPart + tile_m*ldpart + n]
    long ldpart, sPart;
    int crow_mode;
};

template <int BM, int BN, int BK>
__global__ void __launch_bounds__(256, 2) gemm_f64_kernel(GemmP p) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int HK = BK / 2, QK = BK / 4;                    // double2 per tile row, MFMA k-groups
    constexpr int LA = BM * HK / 256, LB = BN * HK / 256;       // double2 loads per thread per tile
    static_assert(LA >= 1 && LB >= 1, "tile too small for 256 threads");
    __shared__ __attribute__((aligned(16))) double As[2][QK][BM][4];
    __shared__ __attribute__((aligned(16))) double Bs[2][QK][BN][4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
    // heavy tiles first: with a lower-triangular A the K range grows with m, with B = T^T with n
    const int tm = (p.kflags & KA_LE_M) ? tilesM - 1 - (int)blockIdx.y : (int)blockIdx.y;
    const int tn = (p.kflags & KB_LE_N) ? tilesN - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int m0 = tm * BM, n0 = tn * BN;
    if (p.lower && n0 > m0 + BM - 1) return;

    int klo = 0, khi = p.K;
    if (p.kflags & KA_LE_M) khi = min(khi, m0 + BM);
    if (p.kflags & KA_GE_M) klo = max(klo, m0);
    if (p.kflags & KB_LE_N) khi = min(khi, n0 + BN);
    if (p.kflags & KB_GE_N) klo = max(klo, n0);
    klo = klo / BK * BK;
    khi = min(p.K, (khi + BK - 1) / BK * BK);
    const int nk = khi > klo ? (khi - klo) / BK : 0;

    const double* __restrict__ A = p.A + (long)blockIdx.z * p.sA;
    const double* __restrict__ B = p.B + (long)blockIdx.z * p.sB;

    d4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    double2 ra[LA], rb[LB];

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int r = 0; r < LA; ++r) {
            const int idx = tid + 256 * r;
            if (!p.a_mc) {
                const int row = idx / HK, k = (idx % HK) * 2;
                ra[r] = (m0 + row < p.M)
                            ? *reinterpret_cast<const double2*>(A + (long)(m0 + row) * p.lda + k0 + k)
                            : double2{0.0, 0.0};
            } else {
                const int k = idx / (BM / 2), m = (idx % (BM / 2)) * 2;
                ra[r] = (m0 + m < p.M)
                            ? *reinterpret_cast<const double2*>(A + (long)(k0 + k) * p.lda + m0 + m)
                            : double2{0.0, 0.0};
            }
        }
#pragma unroll
        for (int r = 0; r < LB; ++r) {
            const int idx = tid + 256 * r;
            if (!p.b_nc) {
                const int row = idx / HK, k = (idx % HK) * 2;
                rb[r] = (n0 + row < p.N)
                            ? *reinterpret_cast<const double2*>(B + (long)(n0 + row) * p.ldb + k0 + k)
                            : double2{0.0, 0.0};
            } else {
                const int k = idx / (BN / 2), n = (idx % (BN / 2)) * 2;
                rb[r] = (n0 + n < p.N)
                            ? *reinterpret_cast<const double2*>(B + (long)(k0 + k) * p.ldb + n0 + n)
                            : double2{0.0, 0.0};
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int r = 0; r < LA; ++r) {
            const int idx = tid + 256 * r;
            if (!p.a_mc) {
                const int row = idx / HK, k = (idx % HK) * 2;
                *reinterpret_cast<double2*>(&As[buf][k >> 2][row][k & 3]) = ra[r];
            } else {
                const int k = idx / (BM / 2), m = (idx % (BM / 2)) * 2;
                As[buf][k >> 2][m][k & 3] = ra[r].x;
                As[buf][k >> 2][m + 1][k & 3] = ra[r].y;
            }
        }
#pragma unroll
        for (int r = 0; r < LB; ++r) {
            const int idx = tid + 256 * r;
            if (!p.b_nc) {
                const int row = idx / HK, k = (idx % HK) * 2;
                *reinterpret_cast<double2*>(&Bs[buf][k >> 2][row][k & 3]) = rb[r];
            } else {
                const int k = idx / (BN / 2), n = (idx % (BN / 2)) * 2;
                Bs[buf][k >> 2][n][k & 3] = rb[r].x;
                Bs[buf][k >> 2][n + 1][k & 3] = rb[r].y;
            }
        }
    };

    if (nk > 0) {
        load_tiles(klo);
        store_tiles(0);
    }
    __syncthreads();
    int cur = 0;
    const int fr = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tiles(klo + (kt + 1) * BK);
#pragma unroll
        for (int q = 0; q < QK; ++q) {
            double a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[cur][q][wm * WM + i * 16 + fr][fk];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[cur][q][wn * WN + j * 16 + fr][fk];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    if (p.epi == EPI_STORE) {
        double* __restrict__ C = p.C + (long)blockIdx.z * p.sC;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * WM + i * 16 + crow(lane, r, p.crow_mode);
                    const int n = n0 + wn * WN + j * 16 + fr;
                    if (m < p.M && n < p.N && (!p.lower || n <= m)) {
                        double* c = C + (long)m * p.ldc + n;
                        double v = p.alpha * acc[i][j][r];
                        if (p.beta != 0.0) v += p.beta * (*c);
                        *c = v;
                    }
                }
    } else {
        // column sums of squares over this block's BM rows (rows >= M hold exact zeros)
        double* red = &As[0][0][0][0];  // [2][BN], free after the final barrier of the K loop
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += acc[i][j][r] * acc[i][j][r];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 16) red[wm * BN + wn * WN + j * 16 + lane] = s;
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N)
            p.part[(long)blockIdx.z * p.sPart + (long)tm * p.ldpart + n0 + tid] = red[tid] + red[BN + tid];
    }
}

// Host-side launcher.  f64 MFMA on gfx950 is issue-latency limited per wave (one
// v_mfma_f64_16x16x4_f64 per ~142 cycles from a single wave, ~47 TFLOP/s chip-wide only with >= 2 waves
// per SIMD -- tools/ubench/mfma_f64_bench.hip), so the tile is chosen to put >= 2 workgroups of 4 waves
// on every CU: 128^2 tiles for the large contractions, 64^2 below that, and 32^2 tiles with a 64-deep K
// step for the small latency-bound products of the factorisation recursion.
// Returns the tile edge used (the caller of EPI_COLSUMSQ sizes `part` with it).
inline int gemm_pick_tile(const GemmP& p, int batch) {
    auto blocks = [&](int t) {
        const long b = (long)((p.M + t - 1) / t) * ((p.N + t - 1) / t) * batch;
        return p.lower ? (b + 1) / 2 : b;
    };
    if (blocks(128) >= 512) return 128;
    if (blocks(64) >= 512) return 64;
    return 32;
}

inline int launch_gemm(const GemmP& p, int batch, hipStream_t stream, int force_tile = 0) {
    const int tile = force_tile ? force_tile : gemm_pick_tile(p, batch);
    dim3 grid((p.N + tile - 1) / tile, (p.M + tile - 1) / tile, batch);
    if (tile == 128) {
        hipLaunchKernelGGL((gemm_f64_kernel<128, 128, 16>), grid, dim3(256), 0, stream, p);
    } else if (tile == 64) {
        hipLaunchKernelGGL((gemm_f64_kernel<64, 64, 16>), grid, dim3(256), 0, stream, p);
    } else if (p.K % 64 == 0) {
        hipLaunchKernelGGL((gemm_f64_kernel<32, 32, 64>), grid, dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((gemm_f64_kernel<32, 32, 16>), grid, dim3(256), 0, stream, p);
    }
    return tile;
}

}  // namespace gpmpc
