// DMA-staged (buffer_load ... lds) 128 x 128 GEMM against the register-staged kernel on the variance-GEMM shape:
// result check (column sums of squares) and timing of ring depths / wave layouts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../gp_mpc_amd/csrc/gemm_f64_dma.hpp"
using namespace gpmpc;

template <class F> float timeit(F f, int reps = 5) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, B = argc > 2 ? atoi(argv[2]) : 10112;
    std::vector<double> hA((size_t)N * N, 0.0), hB((size_t)B * N);
    for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) hA[(size_t)i * N + j] = ((i * 7 + j * 13) % 101 - 50) * 1e-3;   // lower triangular
    for (size_t i = 0; i < hB.size(); ++i) hB[i] = ((i * 31) % 97 - 48) * 1e-2;
    double *A, *Bm, *part, *part2;
    const int tiles = (N + 63) / 64;
    hipMalloc(&A, hA.size() * 8); hipMalloc(&Bm, hB.size() * 8);
    hipMalloc(&part, (size_t)tiles * B * 8); hipMalloc(&part2, (size_t)tiles * B * 8);
    hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice); hipMemcpy(Bm, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
    GemmP v; memset(&v, 0, sizeof(v));
    v.alpha = 1.0; v.A = A; v.lda = N; v.B = Bm; v.ldb = N; v.M = N; v.N = B; v.K = N;
    v.epi = EPI_COLSUMSQ; v.part = part; v.ldpart = B; v.sPart = (long)tiles * B;
    std::vector<double> r0((size_t)tiles * B), r1((size_t)tiles * B);
    for (int tri = 1; tri >= 0; --tri) {
        v.kflags = tri ? KA_LE_M : 0;
        const double fl = tri ? (double)N * (N + 1) * B : 2.0 * N * N * B;
        v.part = part;
        float t = timeit([&] { launch_gemm_cfg<128, 128, 16, 2, 4>(v, 1, 0, 0); });
        printf("%s register-staged 2x4      : %7.3f ms %6.2f TF\n", tri ? "tri  " : "dense", t, fl / t * 1e-9);
        hipMemcpy(r0.data(), part, r0.size() * 8, hipMemcpyDeviceToHost);
        v.part = part2;
#define VARX(BM, BN, WM, WN, S, WPS) { hipMemset(part2, 0, r1.size() * 8); \
        float t = timeit([&] { launch_gemm_dma<BM, BN, WM, WN, S, WPS>(v, 1, 0, 0); }); \
        hipMemcpy(r1.data(), part2, r1.size() * 8, hipMemcpyDeviceToHost); \
        double e = 0, m = 0; /* column sums over all row tiles (tile heights differ between variants) */ \
        for (int c = 0; c < B; ++c) { double a = 0, b = 0; for (int t = 0; t < (N + 127) / 128; ++t) a += r0[(size_t)t * B + c]; \
            for (int t = 0; t < (N + BM - 1) / BM; ++t) b += r1[(size_t)t * B + c]; e = fmax(e, fabs(a - b)); m = fmax(m, fabs(a)); } \
        printf("%s dma tile %dx%d waves %dx%d stages %d wps %d : %7.3f ms %6.2f TF   rel err %.2e\n", tri ? "tri  " : "dense", BM, BN, WM, WN, S, WPS, t, fl / t * 1e-9, e / m); }
        VARX(128, 128, 2, 4, 2, 4)
        VARX(128, 64, 4, 2, 3, 4)
        VARX(128, 64, 4, 2, 2, 4)
        VARX(64, 128, 2, 4, 3, 4)
        VARX(128, 64, 2, 2, 3, 3)
        VARX(64, 64, 2, 2, 3, 4)
    }
    return 0;
}
