// Per-leaf time of leaf_body inside ONE persistent workgroup (as the chain kernel runs it): REPS leaves back to back on
// LDS-resident data, wall-clock stamps around the loop.  Build variants with -DLEAF_HEADER=\"...\" to compare headers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef LEAF_HEADER
#define LEAF_HEADER "../../gp_mpc_amd/csrc/leaf64.hpp"
#endif
#include LEAF_HEADER
using namespace gpmpc;
__global__ void __launch_bounds__(256) loop_kernel(const double* A, double* Lout, double* Iout, long long* stamps, int reps, int phases) {
    __shared__ double S[64 * LS], T[64 * LS], U[64 * LS], Dr[64], S0[64 * LS];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 4096; idx += 256) {
        const int rr = idx >> 6, cc = idx & 63;
        S0[rr * LS + cc] = (cc <= rr) ? A[rr * 64 + cc] : 0.0;
        T[rr * LS + cc] = 0.0;
    }
    __syncthreads();
    int bad = 0;
    const long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        for (int idx = tid; idx < 4096; idx += 256) { const int rr = idx >> 6, cc = idx & 63; S[rr * LS + cc] = S0[rr * LS + cc]; }
        __syncthreads();
        bad += leaf_body(S, T, U, Dr, 1, phases, 0);
    }
    const long long t1 = wall_clock64();
    if (tid == 0) { stamps[0] = t0; stamps[1] = t1; stamps[2] = bad; }
    for (int idx = tid; idx < 4096; idx += 256) { const int rr = idx >> 6, cc = idx & 63; Lout[idx] = S[rr * LS + cc]; Iout[idx] = T[rr * LS + cc]; }
}
int main() {
    const int n = 64;
    std::vector<double> h(n * n), L(n * n), I(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) h[i * n + j] = (i == j ? 70.0 : 0.0) + 1.0 / (1 + abs(i - j));
    double *A, *dL, *dI; long long* st;
    hipMalloc(&A, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dI, n * n * 8); hipMalloc(&st, 64);
    hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice);
    const int masks[] = {0, 1, 3, 7, 15};
    const char* names[] = {"copy only", "+panels", "+rank-16 updates", "+16x16 inverses", "full leaf"};
    for (int v = 0; v < 5; ++v) {
        long long s[3];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(loop_kernel, dim3(1), dim3(256), 0, 0, (const double*)A, dL, dI, st, 200, masks[v]);
            hipDeviceSynchronize();
        }
        hipMemcpy(s, st, 24, hipMemcpyDeviceToHost);
        printf("%-20s %7.2f us per leaf\n", names[v], (s[1] - s[0]) / 100.0 / 200);
    }
    // correctness of the full leaf: L L^T = A, Inv L = I
    hipMemcpy(L.data(), dL, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(I.data(), dI, n * n * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
        double s = 0, t = 0;
        for (int k = 0; k <= j; ++k) s += L[i * 64 + k] * L[j * 64 + k];
        for (int k = j; k <= i; ++k) t += I[i * 64 + k] * L[k * 64 + j];
        e1 = fmax(e1, fabs(s - h[i * n + j])); e2 = fmax(e2, fabs(t - (i == j)));
    }
    printf("max |L L^T - A| = %.2e   max |Inv L - I| = %.2e\n", e1, e2);
    return 0;
}
