// Does the 64 x 64 leaf slow down because it SHARES A CU with trailing-update workgroups, or because the
// memory system is busy?  Time back-to-back leaf launches on stream A while stream B runs (a) nothing,
// (b) an HBM-streaming kernel with 1 small workgroup per CU (memory contention, CUs not full),
// (c) an MFMA-only kernel filling every CU (issue/LDS contention, no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#include "../../gp_mpc_amd/csrc/leaf64.hpp"
using namespace gpmpc;
__global__ void __launch_bounds__(256) k_stream(double* p, long n, int reps) {
    double s = 0;
    for (int r = 0; r < reps; ++r)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { s += p[i]; p[i] = s * 1e-9; }
    if (s == 12345.678) p[0] = s;
}
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters) {
    d4 c0 = d4{0, 0, 0, 0}, c1 = c0;
    const double a = 1.0 + 1e-9 * threadIdx.x;
    for (int i = 0; i < iters; ++i) { c0 = mfma16(a, a, c0); c1 = mfma16(a, a, c1); }
    out[(long)blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1];
}
int main() {
    const int n = 64;
    std::vector<double> h(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) h[i * n + j] = (i == j ? 70.0 : 0.0) + 1.0 / (1 + abs(i - j));
    double *A, *L, *I, *big, *out; int* info;
    hipMalloc(&A, n * n * 8); hipMalloc(&L, n * n * 8); hipMalloc(&I, n * n * 8); hipMalloc(&info, 4);
    const long nbig = 1L << 28;   // 2 GB
    hipMalloc(&big, nbig * 8); hipMemset(big, 0, nbig * 8); hipMalloc(&out, 4096L * 256 * 8);
    hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice); hipMemset(info, 0, 4);
    hipStream_t sa, sb, sm; hipStreamCreate(&sa); hipStreamCreate(&sb);
    {   // sm: stream whose kernels may not use every 8th CU
        hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
        const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
        std::vector<uint32_t> mask(words, 0u);
        for (int i = 0; i < ncu; ++i) if (i % 8 != 7) mask[i / 32] |= (1u << (i % 32));
        hipError_t e = hipExtStreamCreateWithCUMask(&sm, (uint32_t)words, mask.data());
        printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e));
    }
    const char* names[] = {"alone", "with HBM streaming (1 WG/CU)", "with HBM streaming (8 WG/CU)", "with MFMA-only (4 WG/CU)", "with MFMA-only on CU-masked stream", "with MFMA-only (1 WG/CU)"};
    for (int v = 0; v < 6; ++v) {
        if (v == 4) { hipLaunchKernelGGL(k_mfma, dim3(1024), dim3(256), 0, sm, out, 400000); }
        if (v == 5) hipLaunchKernelGGL(k_mfma, dim3(256), dim3(256), 0, sb, out, 400000);
        if (v == 1) hipLaunchKernelGGL(k_stream, dim3(256), dim3(256), 0, sb, big, nbig, 2);
        if (v == 2) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, sb, big, nbig, 6);
        if (v == 3) hipLaunchKernelGGL(k_mfma, dim3(1024), dim3(256), 0, sb, out, 400000);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(leaf64_kernel, dim3(1, 1, 1), dim3(256), 0, sa, (const double*)A, L, I, (long)n, (long)n * n, 0, 1, info, 0, 15);
        hipEventRecord(e0, sa);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(leaf64_kernel, dim3(1, 1, 1), dim3(256), 0, sa, (const double*)A, L, I, (long)n, (long)n * n, 0, 1, info, 0, 15);
        hipEventRecord(e1, sa); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipEvent_t eb; hipEventCreate(&eb); hipEventRecord(eb, sb);
        const bool still = hipEventQuery(eb) != hipSuccess;
        printf("leaf %-34s %7.2f us per launch   (background still running at the end: %s)\n", names[v], ms * 1000 / 200, still ? "yes" : "NO - too short");
        hipStreamSynchronize(sb); hipStreamSynchronize(sm);
    }
    return 0;
}
