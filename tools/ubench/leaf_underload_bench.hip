// The leaf's panels take 1.3 us each in isolation and 2.0 us inside the chain kernel (in-kernel stamps).  Is that the
// rest of the chip being busy?  One persistent workgroup runs leaves back to back (its ~140 KB of LDS keep its CU to
// itself) while a second queue runs, on 224 other CUs (one 512-thread workgroup each, 100 KB of LDS requested):
// nothing / a register-fed fp64 MFMA loop / the same with its operands re-read from LDS / an HBM stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../gp_mpc_amd/csrc/leaf64.hpp"
using namespace gpmpc;
__global__ void __launch_bounds__(256) loop_kernel(const double* A, long long* stamps, int reps, int phases) {
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
}
__global__ void __launch_bounds__(512) k_mfma(double* out, int iters, int use_lds) {
    extern __shared__ double sm[];
    for (int i = threadIdx.x; i < 8192; i += 512) sm[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    d4 c0 = d4{0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = 1.0 + 1e-9 * threadIdx.x, b = a;
    for (int i = 0; i < iters; ++i) {
        if (use_lds) { a = sm[(threadIdx.x + 64 * i) & 8191]; b = sm[(threadIdx.x + 64 * i + 4096) & 8191]; }
        c0 = mfma16(a, b, c0); c1 = mfma16(a, b, c1); c2 = mfma16(b, a, c2); c3 = mfma16(b, a, c3);
    }
    out[(long)blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void __launch_bounds__(512) k_stream(double* p, long n, int reps) {
    extern __shared__ double sm[];
    double s = 0;
    for (int r = 0; r < reps; ++r)
        for (long i = (long)blockIdx.x * 512 + threadIdx.x; i < n; i += (long)gridDim.x * 512) s += p[i];
    if (s == 12345.678) p[0] = s;
}
int main() {
    const int n = 64;
    std::vector<double> h(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) h[i * n + j] = (i == j ? 70.0 : 0.0) + 1.0 / (1 + abs(i - j));
    double *A, *out, *big; long long* st;
    hipMalloc(&A, n * n * 8); hipMalloc(&st, 64); hipMalloc(&out, 256L * 512 * 8);
    const long nbig = 1L << 27;
    hipMalloc(&big, nbig * 8); hipMemset(big, 0, nbig * 8);
    hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    const char* loads[] = {"nothing else", "224 CUs: fp64 MFMA, register operands", "224 CUs: fp64 MFMA, operands from LDS", "224 CUs: HBM stream"};
    const int masks[] = {1, 15};
    const char* what[] = {"copy + panels", "full leaf"};
    for (int ld = 0; ld < 4; ++ld)
        for (int v = 0; v < 2; ++v) {
            long long s[3];
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(loop_kernel, dim3(1), dim3(256), 0, sa, (const double*)A, st, 2000, masks[v]);
                if (ld == 1) hipLaunchKernelGGL(k_mfma, dim3(224), dim3(512), 100 * 1024, sb, out, 120000, 0);   // ~25 ms
                if (ld == 2) hipLaunchKernelGGL(k_mfma, dim3(224), dim3(512), 100 * 1024, sb, out, 100000, 1);
                if (ld == 3) hipLaunchKernelGGL(k_stream, dim3(224), dim3(512), 100 * 1024, sb, big, nbig, 100);
                hipDeviceSynchronize();
            }
            hipMemcpy(s, st, 24, hipMemcpyDeviceToHost);
            printf("%-42s %-14s %7.2f us per leaf\n", loads[ld], what[v], (s[1] - s[0]) / 100.0 / 2000);
        }
    return 0;
}
