// Variance-GEMM micro-benchmark: wave-tile shapes / priorities of gemm_f64_kernel on the headline shape, in the
// triangular (product) form and dense (for comparison with the vendor DGEMM measured by tools/gpu_refresh.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../gp_mpc_amd/csrc/gemm_f64.hpp"
using namespace gpmpc;

template <int BM, int BN, int BK, int WGM, int WGN>
float run(GemmP p, int reps = 5) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch_gemm_cfg<BM, BN, BK, WGM, WGN>(p, 1, 0, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch_gemm_cfg<BM, BN, BK, WGM, WGN>(p, 1, 0, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const int N = 4096, B = 10112;
    std::vector<double> hA((size_t)N * N, 0.0), hB((size_t)B * N);
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) hA[(size_t)i * N + j] = ((i * 7 + j * 13) % 101 - 50) * 1e-3;
    for (size_t i = 0; i < hB.size(); ++i) hB[i] = ((i * 31) % 97 - 48) * 1e-2;
    double *A, *Bm, *part;
    hipMalloc(&A, hA.size() * 8); hipMalloc(&Bm, hB.size() * 8); hipMalloc(&part, (size_t)(N / 32) * B * 8);
    hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice); hipMemcpy(Bm, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
    GemmP v; memset(&v, 0, sizeof(v));
    v.alpha = 1.0; v.A = A; v.lda = N; v.B = Bm; v.ldb = N; v.M = N; v.N = B; v.K = N;
    v.epi = EPI_COLSUMSQ; v.part = part; v.ldpart = B;
    printf("GPMPC_GEMM_PRIO=%d\n", GPMPC_GEMM_PRIO);
#define VAR(BM, BN, BK, WM, WN) { v.sPart = (long)(N / BM) * B; \
        v.kflags = KA_LE_M; float t = run<BM, BN, BK, WM, WN>(v); v.kflags = 0; float d = run<BM, BN, BK, WM, WN>(v); \
        printf("  tile %3dx%3d bk %2d waves %dx%d : tri %7.3f ms %6.2f TF   dense %7.3f ms %6.2f TF\n", BM, BN, BK, WM, WN, \
               t, (double)N * (N + 1) * B / t * 1e-9, d, 2.0 * N * N * B / d * 1e-9); }
    VAR(128, 128, 16, 2, 4)
    VAR(128, 128, 16, 1, 4)
    VAR(128, 128, 16, 4, 1)
    VAR(128, 128, 16, 2, 2)
    VAR(128, 128, 16, 1, 8)
    VAR(256, 128, 16, 2, 4)
    VAR(128, 256, 16, 2, 4)
    return 0;
}
