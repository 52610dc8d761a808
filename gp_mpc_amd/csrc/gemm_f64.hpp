// Batched fp64 GEMM on the MFMA pipe with triangular-aware K ranges and fused epilogues.
//
//   C[b] = alpha * A[b] * B[b] + beta * C[b]          (epi == EPI_STORE)
//   part[b][tile_m][n] = sum_{m in tile_m} (A[b] B[b])[m][n]^2      (epi == EPI_COLSUMSQ)
//
// This one kernel is every dense contraction of the GP hot path:
//   * Cholesky panel  L21 = A21 invL11^T and trailing update A22 -= L21 L21^T   (a4, optimize.py:346)
//   * triangular inverse  invL21 = -invL22 (L21 invL11)                          (a6, optimize.py:489)
//   * K^-1 = invL^T invL                                                         (a6, optimize.py:490)
//   * predictive variance  sum_i (invL Ks)_ij^2                                  (a9, gp_functions.py:122-126)
//
// This is the register-staged kernel; the DMA-staged kernel of gemm_f64_dma.hpp (same GemmP contract, operand
// tiles loaded straight into LDS) has replaced it for the 128 x 128 and 64 x 64 tiles wherever its preconditions
// hold (gemm_dma_supported), and launch_gemm lives there.  This one remains for the 32-row tiles, the skinny
// variance products and as the fallback.
//
// Design for MI355X (measurements: tools/ubench/mfma_issue_bench.hip, profiles/):
//   * v_mfma_f64_16x16x4_f64 holds a SIMD's matrix pipe for 64 cycles and one wave per SIMD can keep it busy
//     with a clean instruction stream; every other instruction of the same wave costs pipe time, so the staging
//     work is spread over many waves: the large tile (128 x 128) is worked by 8 waves (512 threads, <= 128
//     VGPRs: two workgroups = 16 waves per CU), the small tiles by 4 waves with 4-5 workgroups per CU.
//   * operand tiles go global -> registers -> LDS with a one-tile software pipeline (loads of tile t+1
//     in flight while tile t feeds the matrix pipe, one barrier per K step).  LDS holds each operand in
//     MFMA-fragment order [k/4][row][k%4]: the 64 lanes of a fragment read fetch 64 consecutive
//     doubles (512 contiguous bytes, conflict-free).
//   * triangular operands make the K range depend on the tile.  When every workgroup of the launch
//     is resident at once (no back-filling), each workgroup computes a tile AND its mirror so that
//     all workgroups carry the same work; otherwise heavy tiles are issued first.
//   * tile order is plain row-major, heavy rows first; the hardware deals consecutive workgroups to
//     the 8 XCDs round-robin, which spreads the heavy tiles of a triangular product evenly.  (An
//     XCD-contiguous 8 x 8 patch order -- `remap` -- was measured 1.6x SLOWER on the variance GEMM:
//     it hands all heavy rows to one XCD.  It is kept only for the micro-benchmark.)
#pragma once
#include <cstdlib>
#include "mfma_f64.hpp"
#include "wg_sync.hpp"

// Ablation bits for tools/ubench/gemm_ablate_bench.hip only (the results are then wrong; timing study):
// 1 no global tile loads, 2 no LDS tile stores, 4 no barrier in the K loop, 8 no LDS fragment reads.
#ifndef GPMPC_GEMM_ABLATE
#define GPMPC_GEMM_ABLATE 0
#endif
#ifndef GPMPC_GEMM_SPLIT
#define GPMPC_GEMM_SPLIT false
#endif

namespace gpmpc {

// raw buffer resource over [base, base + bytes): loads beyond the range return 0 (gfx9 family word 3: 0x00020000)
#ifdef GPMPC_EMULATED
struct gpmpc_rsrc_t { const char* base; unsigned bytes; };
inline gpmpc_rsrc_t gpmpc_make_rsrc(const void* base, unsigned bytes) { return gpmpc_rsrc_t{(const char*)base, bytes}; }
inline double2 gpmpc_buffer_load_d2(const gpmpc_rsrc_t& r, unsigned voff, int soff) {
    const unsigned long o = (unsigned long)voff + (unsigned long)(unsigned)soff;
    if (o + 16 > r.bytes) return double2{0.0, 0.0};
    return *reinterpret_cast<const double2*>(r.base + o);
}
#else
typedef __amdgpu_buffer_rsrc_t gpmpc_rsrc_t;
__device__ __forceinline__ gpmpc_rsrc_t gpmpc_make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double2 gpmpc_buffer_load_d2(gpmpc_rsrc_t r, unsigned voff, int soff) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
    double2 d;
    __builtin_memcpy(&d, &v, 16);
    return d;
}
#endif


enum { KA_LE_M = 1,   // A(m,k) == 0 for k > m   (A lower triangular)
       KA_GE_M = 2,   // A(m,k) == 0 for k < m   (A = T^T, T lower triangular)
       KB_LE_N = 4,   // B(k,n) == 0 for k > n   (B = T^T, T lower triangular)
       KB_GE_N = 8 }; // B(k,n) == 0 for k < n   (B lower triangular)
enum { EPI_STORE = 0, EPI_COLSUMSQ = 1 };
enum { PAIR_NONE = 0, PAIR_M = 1, PAIR_N = 2 };

struct GemmP {
    const double* A;
    const double* B;
    double* C;
    long lda, ldb, ldc;
    long sA, sB, sC;      // batch strides in elements
    int zdiv;             // > 0: batch index z = z2 * zdiv + z1 with strides (sA, sA2) etc. (nodes x outputs)
    long sA2, sB2, sC2;
    int M, N, K;
    double alpha, beta;
    int a_mc;             // 0: A(m,k) = A[m*lda + k] (K contiguous)   1: A(m,k) = A[k*lda + m]
    int b_nc;             // 0: B(k,n) = B[n*ldb + k] (K contiguous)   1: B(k,n) = B[k*ldb + n]
    int kflags;
    int lower;            // 1: write only n <= m (tiles strictly above the diagonal are skipped)
    int epi;
    double* part;         // EPI_COLSUMSQ: part[b*sPart + tile_m*ldpart + n]
    long ldpart, sPart;
    double* Ct;           // EPI_COLSUMSQ, optional: the product itself, transposed: Ct[b*sCt + n*ldct + m]
    long ldct, sCt;
    int crow_mode;
    int pair;             // set by the launcher
    int tilesMe, tilesNe; // effective tile grid (after pairing), set by the launcher
    int remap;            // 1: XCD-contiguous 8 x 8 patch order, 0: plain row-major tile order
    int npad;             // row-major order: padded width of the tile grid (set by the launcher)
    // hand-offs with the persistent chain kernel (chol_chain.hpp); all null/0 for ordinary launches
    const int* wait_flag; // every workgroup waits for *wait_flag >= 1 before touching its operands
    int* err;             // shared error word (time-out)
    int spin_limit;
    int skip00;           // tile (0,0) belongs to the chain kernel
    int* done_flags;      // tiles (1,0) and (1,1) publish done_flags[0] / done_flags[1] = 1
    long sFlags;          // batch stride of the three flag pointers
    const double* wvec;   // persistent variance product only (vargemm_persist.hpp): w = L^-1 y per matrix (stride sWv) and
    long sWv;             // ... partm[b*sPart + tile_m*ldpart + n] = sum_{m in tile_m} (A B)[m][n] w[m]: the predictive mean ks^T alpha = (L^-1 ks)^T w
    double* partm;
    const int* zmap;      // optional: the launch's matrices are zmap[0 .. batch) of the workspace (a subset); not with flags / part / Ct
};

// the accumulators of one wave, written transposed (EPI_COLSUMSQ with Ct: the few-column products of the sensitivities)
template <int TM, int TN>
__device__ __forceinline__ void gemm_store_transposed(const GemmP& p, const d4 (&acc)[TM][TN], int mw, int nw, int lane) {
    double* __restrict__ Ct = p.Ct + (long)blockIdx.z * p.sCt;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mw + i * 16 + crow(lane, r, p.crow_mode), n = nw + j * 16 + (lane & 15);
                if (m < p.M && n < p.N) Ct[(long)n * p.ldct + m] = acc[i][j][r];
            }
}

template <int BM, int BN, int BK, int WGM, int WGN, bool AMC, bool BNC, bool SPLIT = GPMPC_GEMM_SPLIT, bool BUF = false>
__global__ void __launch_bounds__(64 * WGM * WGN, WGM * WGN / 2) gemm_f64_kernel(GemmP p) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
    constexpr int HK = BK / 2, QK = BK / 4;                    // double2 per tile row, MFMA k-groups
    constexpr int LA = BM * HK / NT, LB = BN * HK / NT;         // double2 loads per thread per tile
    static_assert(LA >= 1 && LB >= 1 && TM >= 1 && TN >= 1, "bad tile configuration");
    __shared__ __attribute__((aligned(16))) double As[2][QK][BM][4];
    __shared__ __attribute__((aligned(16))) double Bs[2][QK][BN][4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;

    // ---- workgroup id -> tile: XCD-contiguous runs of 8 x 8 patches
    int tme, tne;
    if (p.remap) {
        const int nb = (int)gridDim.x, b = (int)blockIdx.x;
        const int xcd = b & 7, q = nb >> 3, r = nb & 7;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        const int pcols = (p.tilesNe + 7) >> 3;
        const int prow = lin / (64 * pcols), rem = lin % (64 * pcols);
        tme = prow * 8 + ((rem & 63) >> 3);
        tne = (rem >> 6) * 8 + (rem & 7);
    } else {
        // row-major tile order, heavy rows first; optionally over a grid whose width is padded to a
        // multiple of 8 so that XCD x (which receives workgroups x, x+8, ...) always works on tile
        // columns = x (mod 8) and its concurrent workgroups share operand panels in its private L2.
        const int wpad = p.npad;
        tme = (int)blockIdx.x / wpad;
        tne = (int)blockIdx.x % wpad;
    }
    if (tme >= p.tilesMe || tne >= p.tilesNe) return;
    if (p.wait_flag) {
        __shared__ int wslot;
        if (!wg_wait2(p.wait_flag + (long)blockIdx.z * p.sFlags, 1, nullptr, 0, p.err + (long)blockIdx.z * p.sFlags,
                      p.spin_limit, &wslot))
            return;
    }

    const int z1 = p.zdiv > 0 ? (int)blockIdx.z % p.zdiv : (p.zmap ? p.zmap[blockIdx.z] : (int)blockIdx.z);
    const int z2 = p.zdiv > 0 ? (p.zmap ? p.zmap[(int)blockIdx.z / p.zdiv] : (int)blockIdx.z / p.zdiv) : 0;
    const double* __restrict__ A = p.A + (long)z1 * p.sA + (long)z2 * p.sA2;
    const double* __restrict__ B = p.B + (long)z1 * p.sB + (long)z2 * p.sB2;
    const int fr = lane & 15, fk = lane >> 4;
    // BUF (both operands K-contiguous, < 4 GB each): tile loads are buffer_load_dwordx4 through a resource whose
    // range check returns zeros for rows >= M / N -- no exec-mask branches, no 64-bit address arithmetic and no
    // zero-fill moves in the main loop (each VALU instruction there costs matrix-pipe issue time, measured with
    // tools/ubench/mfma_issue_bench.hip: +8 ns per MFMA for one v_fma per MFMA); the per-iteration K offset rides
    // in the instruction's scalar offset.
    gpmpc_rsrc_t rsA, rsB;
    if (BUF) {
        rsA = gpmpc_make_rsrc(A, (unsigned)((long)p.M * p.lda * 8));
        rsB = gpmpc_make_rsrc(B, (unsigned)((long)p.N * p.ldb * 8));
    }

    for (int pass = 0; pass < 2; ++pass) {
        // heavy tiles first: with a lower-triangular A the K range grows with m, with B = T^T with n
        int tm = (p.kflags & KA_LE_M) ? tilesM - 1 - tme : tme;
        int tn = (p.kflags & KB_LE_N) ? tilesN - 1 - tne : tne;
        if (pass == 1) {
            if (p.pair == PAIR_M) {
                if (tilesM - 1 - tm == tm) break;
                tm = tilesM - 1 - tm;
            } else if (p.pair == PAIR_N) {
                if (tilesN - 1 - tn == tn) break;
                tn = tilesN - 1 - tn;
            } else {
                break;
            }
        }
        const int m0 = tm * BM, n0 = tn * BN;
        if (p.lower && n0 > m0 + BM - 1) continue;
        if (p.skip00 && tm == 0 && tn == 0) continue;

        int klo = 0, khi = p.K;
        if (p.kflags & KA_LE_M) khi = min(khi, m0 + BM);
        if (p.kflags & KA_GE_M) klo = max(klo, m0);
        if (p.kflags & KB_LE_N) khi = min(khi, n0 + BN);
        if (p.kflags & KB_GE_N) klo = max(klo, n0);
        klo = klo / BK * BK;
        khi = min(p.K, (khi + BK - 1) / BK * BK);
        const int nk = khi > klo ? (khi - klo) / BK : 0;

        d4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

        double2 ra[LA], rb[LB];
        unsigned voA[LA], voB[LB];      // BUF: byte offset of this thread's r-th 16-byte piece at k0 = 0
        if (BUF) {
#pragma unroll
            for (int r = 0; r < LA; ++r) {
                const int idx = tid + NT * r, row = idx / HK, k = (idx % HK) * 2;
                voA[r] = (unsigned)(((long)(m0 + row) * p.lda + k) * 8);
            }
#pragma unroll
            for (int r = 0; r < LB; ++r) {
                const int idx = tid + NT * r, row = idx / HK, k = (idx % HK) * 2;
                voB[r] = (unsigned)(((long)(n0 + row) * p.ldb + k) * 8);
            }
        }

        auto load_tiles = [&](int k0) {
            if (GPMPC_GEMM_ABLATE & 1) return;
            if (BUF) {
#pragma unroll
                for (int r = 0; r < LA; ++r) ra[r] = gpmpc_buffer_load_d2(rsA, voA[r], k0 * 8);
#pragma unroll
                for (int r = 0; r < LB; ++r) rb[r] = gpmpc_buffer_load_d2(rsB, voB[r], k0 * 8);
                return;
            }
#pragma unroll
            for (int r = 0; r < LA; ++r) {
                const int idx = tid + NT * r;
                if (!AMC) {
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
                const int idx = tid + NT * r;
                if (!BNC) {
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
            if (GPMPC_GEMM_ABLATE & 2) return;
#pragma unroll
            for (int r = 0; r < LA; ++r) {
                const int idx = tid + NT * r;
                if (!AMC) {
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
                const int idx = tid + NT * r;
                if (!BNC) {
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
        auto mma_groups = [&](int q0, int q1) {
#pragma unroll
            for (int q = q0; q < q1; ++q) {
                double a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = (GPMPC_GEMM_ABLATE & 8) ? 1.0 + i + q + lane : As[cur][q][wm * WM + i * 16 + fr][fk];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = (GPMPC_GEMM_ABLATE & 8) ? 2.0 + j + q + lane : Bs[cur][q][wn * WN + j * 16 + fr][fk];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
            }
        };
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tiles(klo + (kt + 1) * BK);
            // the staging write of tile kt+1 sits between the two halves of the MFMA work so that the
            // LDS write and the global-load wait are covered by queued matrix instructions
            mma_groups(0, SPLIT ? QK / 2 : QK);
            if (kt + 1 < nk) store_tiles(cur ^ 1);
            if (SPLIT) mma_groups(QK / 2, QK);
            if (!(GPMPC_GEMM_ABLATE & 4)) __syncthreads();
            cur ^= 1;
        }

        if (p.epi == EPI_STORE) {
            double* __restrict__ C = p.C + (long)z1 * p.sC + (long)z2 * p.sC2;
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
            // column sums of squares over this tile's BM rows (rows >= M hold exact zeros)
            double* red = &As[0][0][0][0];  // [WGM][BN], free after the final barrier of the K loop
            if (p.Ct) gemm_store_transposed<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane);
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
            if (tid < BN && n0 + tid < p.N) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < WGM; ++w) t += red[w * BN + tid];
                p.part[(long)blockIdx.z * p.sPart + (long)tm * p.ldpart + n0 + tid] = t;
            }
        }
        if (p.done_flags && tm == 1 && tn <= 1) wg_publish(p.done_flags + (long)blockIdx.z * p.sFlags + tn, 1);
        __syncthreads();  // LDS (tiles / red) is reused by the next pass
    }
}

// ---- host side -------------------------------------------------------------------------------------
struct GemmCfg { int tile; int resident; };   // tile edge, workgroups resident on the whole chip

// Tile choice: >= 2 rounds of resident workgroups for the big tile, else smaller tiles (more
// workgroups for the latency-bound products of the factorisation recursion).
inline int gemm_pick_tile(const GemmP& p, int batch) {
    auto blocks = [&](int t) {
        const long b = (long)((p.M + t - 1) / t) * ((p.N + t - 1) / t) * batch;
        return p.lower ? (b + 1) / 2 : b;
    };
    static const int t128 = getenv("GPMPC_T128") ? atoi(getenv("GPMPC_T128")) : 512;
    // (r03: 256 instead of 512: -14 us on the C2 inverse tail with the DMA-staged 64-row kernel; r04, with the prediction running
    //  behind the tail: 384 / 512 instead of 256: -35 / -30 us per C2 step, C3 and C4 unchanged -- tools/gpu_r04_s.sh)
    static const int t64 = getenv("GPMPC_T64") ? atoi(getenv("GPMPC_T64")) : 384;
    if (p.N <= 32 || p.M <= 32) return 32;     // skinny products (a handful of prediction points)
    // A triangular operand makes the heaviest tile K / 128 slabs long while the average is half that: unless the
    // average work per workgroup slot (512 of them) reaches the heaviest tile, the heaviest tiles alone set the time
    // and smaller tiles balance better (N = 8192, Ny = 6, B = 256 variance product: 2.42 ms with 128-row tiles, 1.74 with 64).
    bool balanced = true;
    if (p.kflags == KA_LE_M || p.kflags == KA_GE_M || p.kflags == KB_LE_N || p.kflags == KB_GE_N) {
        const bool onM = p.kflags == KA_LE_M || p.kflags == KA_GE_M;
        const long tT = ((onM ? p.M : p.N) + 127) / 128, tO = ((onM ? p.N : p.M) + 127) / 128;
        balanced = (double)(tO * batch) * (tT + 1) / 2.0 / 512.0 >= 1.0 || tT <= 2;   // (average per slot) / (heaviest tile)
    }
    if (p.lower && p.kflags == (KA_GE_M | KB_GE_N)) {
        // K^-1 = L^-T L^-1: tile (tm, tn <= tm) sums over k >= 128 tm, so tile (0, 0) is K / 16 slabs long while the
        // average workgroup slot gets sum_tm (tm + 1)(T - tm) / 512 of them: at N = 4096 the one heaviest 128-row tile IS
        // the duration of the launch (1.03 ms), with 64-row tiles four times as many slots share the long rows
        const long T = (p.M + 127) / 128;
        double total = 0.0;
        for (long tm = 0; tm < T; ++tm) total += (double)(tm + 1) * (double)(T - tm);
        balanced = total * batch / 512.0 >= 2.0 * (double)T;
    }
    if (blocks(128) >= t128 && balanced) return 128;
    if (blocks(64) >= t64) return 64;
    return 32;
}

template <int BM, int BN, int BK, int WGM, int WGN>
inline void launch_gemm_cfg(GemmP p, int batch, hipStream_t stream, int resident, int min_pair_blocks = 512) {
    // operand orientations are compile-time (fewer registers and no branches in the staging code)
    const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
    p.pair = PAIR_NONE;
    p.tilesMe = tilesM;
    p.tilesNe = tilesN;
    // balanced pairing only when the whole launch is co-resident (no back-filling possible)
    const bool one_flag = (p.kflags == KA_LE_M || p.kflags == KA_GE_M || p.kflags == KB_LE_N || p.kflags == KB_GE_N);
    const long nblocks = (long)tilesM * tilesN * batch;
    if (one_flag && !p.lower && nblocks <= (long)resident && nblocks >= min_pair_blocks) {   // small launches want parallelism, not balance
        if ((p.kflags & (KA_LE_M | KA_GE_M)) && tilesM >= 2) {
            p.pair = PAIR_M;
            p.tilesMe = (tilesM + 1) / 2;
        } else if ((p.kflags & (KB_LE_N | KB_GE_N)) && tilesN >= 2) {
            p.pair = PAIR_N;
            p.tilesNe = (tilesN + 1) / 2;
        }
    }
    const int prows = (p.tilesMe + 7) / 8, pcols = (p.tilesNe + 7) / 8;
    // XCD-consistent column residues (pad the grid width to a multiple of 8) cut the fetched bytes of the
    // variance GEMM by 27 % (3.33 -> 2.43 GB raw FETCH_SIZE) but cost 1-3 % time on every shape measured
    // (the kernels are MFMA-issue bound, not L2 bound): off by default, GPMPC_PAD_MIN=<tiles> enables it.
    static const int pad_min = getenv("GPMPC_PAD_MIN") ? atoi(getenv("GPMPC_PAD_MIN")) : (1 << 30);
    p.npad = (p.tilesNe >= pad_min) ? ((p.tilesNe + 7) & ~7) : p.tilesNe;
    dim3 grid(p.remap ? prows * pcols * 64 : p.tilesMe * p.npad, 1, batch);
    const dim3 block(64 * WGM * WGN);
    static const bool use_buf = !(getenv("GPMPC_GEMM_BUF") && atoi(getenv("GPMPC_GEMM_BUF")) == 0);
    const bool small = (long)p.M * p.lda * 8 < (1L << 32) && (long)p.N * p.ldb * 8 < (1L << 32);
    if (!p.a_mc && !p.b_nc && use_buf && small)
        hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, BK, WGM, WGN, false, false, GPMPC_GEMM_SPLIT, true>), grid, block, 0,
                           stream, p);
    else if (!p.a_mc && !p.b_nc)
        hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, BK, WGM, WGN, false, false>), grid, block, 0, stream, p);
    else if (!p.a_mc && p.b_nc)
        hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, BK, WGM, WGN, false, true>), grid, block, 0, stream, p);
    else if (p.a_mc && !p.b_nc)
        hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, BK, WGM, WGN, true, false>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, BK, WGM, WGN, true, true>), grid, block, 0, stream, p);
}

}  // namespace gpmpc
