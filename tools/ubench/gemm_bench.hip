// GEMM-core micro-benchmark: tile shape x waves x scheduling on the shapes of the hot path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../gp_mpc_amd/csrc/gemm_f64.hpp"
using namespace gpmpc;

template <int BM, int BN, int BK, int WGM, int WGN>
float run(GemmP p, int batch, int resident, int remap, int allow_pair, int reps = 5) {
    p.remap = remap;
    if (!allow_pair) resident = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch_gemm_cfg<BM, BN, BK, WGM, WGN>(p, batch, 0, resident);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch_gemm_cfg<BM, BN, BK, WGM, WGN>(p, batch, 0, resident);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const int N = 4096, B = 10112;
    std::vector<double> hA((size_t)N * N, 0.0), hB((size_t)B * N);
    for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) hA[(size_t)i * N + j] = ((i * 7 + j * 13) % 101 - 50) * 1e-3;
    for (size_t i = 0; i < hB.size(); ++i) hB[i] = ((i * 31) % 97 - 48) * 1e-2;
    double *A, *Bm, *C, *part;
    hipMalloc(&A, hA.size() * 8); hipMalloc(&Bm, hB.size() * 8); hipMalloc(&C, (size_t)N * N * 8); hipMalloc(&part, (size_t)(N / 32) * B * 8);
    hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice); hipMemcpy(Bm, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
    GemmP v; memset(&v, 0, sizeof(v));
    v.alpha = 1.0; v.A = A; v.lda = N; v.B = Bm; v.ldb = N; v.M = N; v.N = B; v.K = N; v.kflags = KA_LE_M;
    v.epi = EPI_COLSUMSQ; v.part = part; v.ldpart = B;
    const double vf = (double)N * (N + 1) * B;
    printf("== variance GEMM  M=%d N=%d K=%d (tri A), flops %.3e\n", N, B, N, vf);
#define VAR(BM, BN, BK, WM, WN, RES) for (int rm = 0; rm < 2; ++rm) { v.sPart = (long)(N / BM) * B; float ms = run<BM, BN, BK, WM, WN>(v, 1, RES, rm, 0); \
        printf("  tile %3dx%3d bk %2d waves %dx%d remap %d : %8.3f ms  %6.2f TF\n", BM, BN, BK, WM, WN, rm, ms, vf / ms * 1e-9); }
    VAR(128, 128, 16, 2, 2, 512)
    VAR(128, 128, 16, 2, 4, 512)
    VAR(128, 128, 16, 4, 2, 512)
    VAR(128, 128, 16, 4, 4, 512)
    VAR(256, 128, 16, 4, 4, 256)
    VAR(128, 256, 16, 4, 4, 256)
    //VAR(128, 64, 16, 2, 2, 768)
    //VAR(64,128,16,2,2,768)
    VAR(64, 64, 16, 2, 2, 1024)
    VAR(128, 128, 32, 2, 4, 512)
    // L6-type products of the factorisation: 2048^3 with triangular B (L21 = A21 inv11^T), SYRK lower, tri A
    const int H = 2048;
    GemmP g; memset(&g, 0, sizeof(g));
    g.alpha = 1.0; g.A = A; g.lda = N; g.B = A; g.ldb = N; g.C = C; g.ldc = N; g.M = H; g.N = H; g.K = H;
    const double gf = (double)H * H * H;   // triangular: half of 2 H^3
#define L6(NAME, KF, LOW, BM, BN, BK, WM, WN, RES) for (int rm = 0; rm < 2; ++rm) for (int pr = 0; pr < 2; ++pr) { g.kflags = KF; g.lower = LOW; \
        float ms = run<BM, BN, BK, WM, WN>(g, 1, RES, rm, pr); \
        printf("  %-10s tile %3dx%3d bk %2d waves %dx%d remap %d pair %d : %8.3f ms  %6.2f TF\n", NAME, BM, BN, BK, WM, WN, rm, pr, ms, gf / ms * 1e-9); }
    printf("== 2048^3 factorisation products (h^3 useful flops each)\n");
    L6("triB", KB_LE_N, 0, 64, 64, 16, 2, 2, 1024)
    L6("triB", KB_LE_N, 0, 128, 128, 16, 2, 4, 512)
    L6("triB", KB_LE_N, 0, 128, 128, 16, 2, 2, 512)
    L6("triB", KB_LE_N, 0, 32, 32, 32, 2, 2, 99999)
    L6("syrk", 0, 1, 64, 64, 16, 2, 2, 1024)
    L6("syrk", 0, 1, 128, 128, 16, 2, 4, 512)
    L6("triA", KA_LE_M, 0, 64, 64, 16, 2, 2, 1024)
    const int H2 = 1024; g.M = g.N = g.K = H2;
    const double gf2 = (double)H2 * H2 * H2;
#define L5(NAME, KF, LOW, BM, BN, BK, WM, WN, RES) for (int pr = 0; pr < 2; ++pr) { g.kflags = KF; g.lower = LOW; \
        float ms = run<BM, BN, BK, WM, WN>(g, 1, RES, 1, pr); \
        printf("  %-10s tile %3dx%3d bk %2d waves %dx%d pair %d : %8.3f ms  %6.2f TF\n", NAME, BM, BN, BK, WM, WN, pr, ms, gf2 / ms * 1e-9); }
    printf("== 1024^3 products\n");
    L5("triB", KB_LE_N, 0, 64, 64, 16, 2, 2, 1024)
    L5("triB", KB_LE_N, 0, 32, 32, 32, 2, 2, 99999)
    L5("syrk", 0, 1, 32, 32, 32, 2, 2, 99999)
    L5("syrk", 0, 1, 64, 64, 16, 2, 2, 99999)
    return 0;
}
