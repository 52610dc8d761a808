// Experiment: fp64 MFMA GEMM for K-contiguous operands WITHOUT LDS or barriers -- every lane loads its
// own MFMA fragments straight from global memory (32 B per lane = 4 consecutive k of one row, the k order
// inside a 16-chunk is permuted identically for A and B, which a sum over k does not care about).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// C tile per wave: (16 TM) x (16 TN); block = WGM x WGN waves.  A[m][k] = A[m*lda+k], B[k][n] = B[n*ldb+k].
// tri: K range [0, m0 + BM).  Output: column sums of squares -> atomicAdd-free partial per (tile_m, n).
template <int TM, int TN, int WGM, int WGN, int PF>
__global__ void __launch_bounds__(64 * WGM * WGN, (WGM * WGN >= 8 ? 4 : 2)) gemm_direct(const double* __restrict__ A, const double* __restrict__ B,
                                                           double* __restrict__ part, int M, int N, int K, long lda, long ldb, long ldp) {
    constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WGN, wn = wave % WGN;
    const int tilesN = (N + BN - 1) / BN, tilesM = (M + BM - 1) / BM;
    const int tm = tilesM - 1 - (int)blockIdx.x / tilesN, tn = (int)blockIdx.x % tilesN;
    const int m0 = tm * BM + wm * 16 * TM, n0 = tn * BN + wn * 16 * TN;
    const int khi = min(K, tm * BM + BM);
    const int fr = lane & 15, fk = lane >> 4;
    const double* ap[TM];
    const double* bp[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) ap[i] = A + (long)(m0 + 16 * i + fr) * lda + 4 * fk;
#pragma unroll
    for (int j = 0; j < TN; ++j) bp[j] = B + (long)min(n0 + 16 * j + fr, N - 1) * ldb + 4 * fk;
    d4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = d4{0, 0, 0, 0};
    d4 ra[PF][TM], rb[PF][TN];
    auto load = [&](int slot, int k0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) ra[slot][i] = *reinterpret_cast<const d4*>(ap[i] + k0);
#pragma unroll
        for (int j = 0; j < TN; ++j) rb[slot][j] = *reinterpret_cast<const d4*>(bp[j] + k0);
    };
    const int nk = khi / 16;
#pragma unroll
    for (int s = 0; s < PF - 1; ++s) if (s < nk) load(s, 16 * s);
    for (int kt = 0; kt < nk; kt += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int c = kt + u;
            if (c < nk) {
                if (c + PF - 1 < nk) load((u + PF - 1) % PF, 16 * (c + PF - 1));
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[u][i][s], rb[u][j][s], acc[i][j], 0, 0, 0);
            }
        }
    }
    // epilogue: column sums of squares of this wave's rows (one value per column per wave)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        double s = 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][j][r] * acc[i][j][r];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const int n = n0 + 16 * j + lane;
        if (lane < 16 && n < N) part[(long)(tm * WGM + wm) * ldp + n] = s;
    }
}

template <int TM, int TN, int WGM, int WGN, int PF>
void run(const char* name, const double* A, const double* B, double* part, int M, int N, int K) {
    constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN;
    const int blocks = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() { hipLaunchKernelGGL((gemm_direct<TM, TN, WGM, WGN, PF>), dim3(blocks), dim3(64 * WGM * WGN), 0, 0, A, B, part, M, N, K, (long)K, (long)K, (long)N); };
    go();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) go();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double fl = (double)M * (M + 1) * N;
    printf("  direct %-28s tile %3dx%3d waves %dx%d pf %d : %8.3f ms  %6.2f TF  (err %s)\n", name, BM, BN, WGM, WGN, PF, ms, fl / ms * 1e-9, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int N = 4096, B = 10112;
    std::vector<double> hA((size_t)N * N, 0.0), hB((size_t)B * N);
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double)(x >> 11) / 9007199254740992.0 - 0.5; };
    for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) hA[(size_t)i * N + j] = rnd();
    for (size_t i = 0; i < hB.size(); ++i) hB[i] = rnd();
    double *A, *Bm, *part;
    hipMalloc(&A, hA.size() * 8); hipMalloc(&Bm, hB.size() * 8); hipMalloc(&part, (size_t)(N / 16) * B * 8);
    hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice); hipMemcpy(Bm, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
    printf("== variance GEMM shape M=%d N=%d K=%d (random data)\n", N, B, N);
    run<4, 2, 2, 4, 2>("64x32 per wave", A, Bm, part, N, B, N);
    run<4, 2, 2, 4, 3>("64x32 per wave", A, Bm, part, N, B, N);
    run<2, 2, 4, 4, 2>("32x32 per wave, 16 waves", A, Bm, part, N, B, N);
    run<2, 2, 4, 4, 3>("32x32 per wave, 16 waves", A, Bm, part, N, B, N);
    run<2, 2, 2, 2, 2>("32x32 per wave, 4 waves", A, Bm, part, N, B, N);
    run<2, 2, 2, 2, 4>("32x32 per wave, 4 waves", A, Bm, part, N, B, N);
    run<4, 4, 2, 2, 2>("64x64 per wave, 4 waves", A, Bm, part, N, B, N);
    run<2, 4, 2, 2, 2>("32x64 per wave, 4 waves", A, Bm, part, N, B, N);
    run<2, 2, 2, 4, 2>("32x32 per wave, 8 waves", A, Bm, part, N, B, N);
    run<2, 2, 2, 4, 4>("32x32 per wave, 8 waves", A, Bm, part, N, B, N);
    return 0;
}
