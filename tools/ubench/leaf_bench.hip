// Phase timing of the 64 x 64 leaf (Cholesky + inverse of a diagonal block): 200 dependent launches per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../gp_mpc_amd/csrc/leaf64.hpp"
using namespace gpmpc;
int main() {
    const int n = 64;
    std::vector<double> h(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) h[i * n + j] = (i == j ? 70.0 : 0.0) + 1.0 / (1 + abs(i - j));
    double *A, *L, *I; int* info;
    hipMalloc(&A, n * n * 8); hipMalloc(&L, n * n * 8); hipMalloc(&I, n * n * 8); hipMalloc(&info, 4);
    hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice); hipMemset(info, 0, 4);
    const char* names[] = {"io only", "+panel", "+panel+update", "+panel+update+inv16", "full (15)", "panel only+asm(9)"};
    const int masks[] = {0, 1, 3, 7, 15, 9};
    for (int v = 0; v < 6; ++v) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(leaf64_kernel, dim3(1, 1, 1), dim3(256), 0, 0, (const double*)A, L, I, (long)n, (long)n * n, 0, 1, info, 0, masks[v]);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(leaf64_kernel, dim3(1, 1, 1), dim3(256), 0, 0, (const double*)A, L, I, (long)n, (long)n * n, 0, 1, info, 0, masks[v]);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-24s %7.2f us per launch (back-to-back, incl. launch gap)\n", names[v], ms * 1000 / 200);
    }
    return 0;
}
