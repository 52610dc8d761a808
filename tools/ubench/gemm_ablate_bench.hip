// Where does the variance GEMM lose matrix-pipe time?  The production kernel with parts of its K loop removed
// (GPMPC_GEMM_ABLATE bits: 1 global loads, 2 LDS stores, 4 barrier, 8 LDS fragment reads); dense and triangular.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../gp_mpc_amd/csrc/gemm_f64.hpp"
using namespace gpmpc;
int main() {
    const int N = 4096, B = 10112;
    std::vector<double> hA((size_t)N * N), hB((size_t)B * N);
    unsigned s = 1;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = (double)(s >> 8) / (1 << 24) - 0.5; }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = (double)(s >> 8) / (1 << 24) - 0.5; }
    double *A, *Bm, *part;
    hipMalloc(&A, hA.size() * 8); hipMalloc(&Bm, hB.size() * 8); hipMalloc(&part, (size_t)(N / 32) * B * 8);
    hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice); hipMemcpy(Bm, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
    GemmP v; memset(&v, 0, sizeof(v));
    v.alpha = 1.0; v.A = A; v.lda = N; v.B = Bm; v.ldb = N; v.M = N; v.N = B; v.K = N;
    v.epi = EPI_COLSUMSQ; v.part = part; v.ldpart = B; v.sPart = (long)(N / 128) * B;
    for (int tri = 0; tri < 2; ++tri) {
        v.kflags = tri ? KA_LE_M : 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        launch_gemm_cfg<128, 128, 16, 2, 4>(v, 1, 0, 0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 5; ++i) launch_gemm_cfg<128, 128, 16, 2, 4>(v, 1, 0, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double fl = tri ? (double)N * (N + 1) * B : 2.0 * N * N * B;
        printf("ablate %2d %s : %7.3f ms  %6.2f TFLOP/s\n", GPMPC_GEMM_ABLATE, tri ? "tri  " : "dense", ms, fl / ms * 1e-9);
    }
    return 0;
}
