// Does the 64 x 64 leaf slow down when its code is not in the instruction cache?  (44 KB of straight-line code; the chain
// kernel runs it once per panel step between ~12 KB of other code, on a CU that shares its 64 KB instruction cache with
// a neighbour running a different kernel.)  The leaf is timed with stamps of its own; between two leaves the workgroup
// runs `junk` KB of s_nop, which evicts that much of the cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../gp_mpc_amd/csrc/leaf64.hpp"
using namespace gpmpc;
template <int I> __device__ __attribute__((noinline)) void junk16() { asm volatile(".rept 4096\n s_nop 0\n .endr" ::: "memory"); }   // 16 KB each, distinct copies
__global__ void __launch_bounds__(256) loop_kernel(const double* A, long long* stamps, int reps, int junk) {
    __shared__ double S[64 * LS], T[64 * LS], U[64 * LS], Dr[64], S0[64 * LS];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 4096; idx += 256) {
        const int rr = idx >> 6, cc = idx & 63;
        S0[rr * LS + cc] = (cc <= rr) ? A[rr * 64 + cc] : 0.0;
        T[rr * LS + cc] = 0.0;
    }
    __syncthreads();
    int bad = 0;
    long long acc = 0;
    for (int r = 0; r < reps; ++r) {
        for (int idx = tid; idx < 4096; idx += 256) { const int rr = idx >> 6, cc = idx & 63; S[rr * LS + cc] = S0[rr * LS + cc]; }
        __syncthreads();
        const long long t0 = wall_clock64();
        bad += leaf_body(S, T, U, Dr, 1, 15, 0);
        __syncthreads();
        acc += wall_clock64() - t0;
        if (junk >= 1) junk16<1>();
        if (junk >= 2) junk16<2>();
        if (junk >= 3) junk16<3>();
        if (junk >= 4) junk16<4>();
        if (junk >= 5) junk16<5>();
        if (junk >= 6) junk16<6>();
        __syncthreads();
    }
    if (tid == 0) { stamps[0] = acc; stamps[1] = bad; }
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
    double* A; long long* st;
    hipMalloc(&A, n * n * 8); hipMalloc(&st, 64);
    hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice);
    const long nbig = 1L << 27;
    double* big; hipMalloc(&big, nbig * 8); hipMemset(big, 0, nbig * 8);
    hipFuncSetAttribute((const void*)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    for (int load = 0; load < 2; ++load)
    for (int junk = 0; junk <= 6; junk += 2) {
        long long s[2];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(loop_kernel, dim3(1), dim3(256), 0, sa, (const double*)A, st, 2000, junk);
            if (load) hipLaunchKernelGGL(k_stream, dim3(224), dim3(512), 100 * 1024, sb, big, nbig, 150);   // HBM stream on 224 other CUs
            hipDeviceSynchronize();
        }
        hipMemcpy(s, st, 16, hipMemcpyDeviceToHost);
        printf("%s, %2d KB of other code between two leaves: %7.2f us per leaf\n", load ? "224 CUs stream HBM" : "chip idle         ", 16 * junk, s[0] / 100.0 / 2000);
    }
    return 0;
}
