// 64 x 16 panel factorisation (the sequential core of the 64 x 64 leaf): v_readlane form (panel_potrf) against the two DPP
// forms (panel_potrf_dpp<., 1> mov_dpp + fma, <., 2> fmac_dpp), one wave, REPS x 4 panels back to back on LDS data; plus
// the issue rate of dependent-free v_fma_f64 / v_fmac_f64_dpp / v_readlane+fma streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../gp_mpc_amd/csrc/leaf64.hpp"
using namespace gpmpc;
template <int V>
__global__ void __launch_bounds__(64) panel_kernel(const double* A, double* Lout, long long* stamps, int reps) {
    __shared__ double S[64 * LS], S0[64 * LS], Dr[64];
    const int lane = threadIdx.x;
    for (int idx = lane; idx < 4096; idx += 64) { const int rr = idx >> 6, cc = idx & 63; S0[rr * LS + cc] = A[rr * 64 + cc]; }
    __syncthreads();
    int bad = 0;
    const long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        for (int idx = lane; idx < 4096; idx += 64) { const int rr = idx >> 6, cc = idx & 63; S[rr * LS + cc] = S0[rr * LS + cc]; }
        __syncthreads();
        if (V == 0) { bad += panel_potrf<0>(S, Dr, lane); bad += panel_potrf<1>(S, Dr, lane); bad += panel_potrf<2>(S, Dr, lane); bad += panel_potrf<3>(S, Dr, lane); }
        if (V == 1) { bad += panel_potrf_dpp<0, 1>(S, Dr, lane); bad += panel_potrf_dpp<1, 1>(S, Dr, lane); bad += panel_potrf_dpp<2, 1>(S, Dr, lane); bad += panel_potrf_dpp<3, 1>(S, Dr, lane); }
        if (V == 2) { bad += panel_potrf_dpp<0, 2>(S, Dr, lane); bad += panel_potrf_dpp<1, 2>(S, Dr, lane); bad += panel_potrf_dpp<2, 2>(S, Dr, lane); bad += panel_potrf_dpp<3, 2>(S, Dr, lane); }
        if (V == 3) { }   // copy only
        __syncthreads();
    }
    const long long t1 = wall_clock64();
    if (lane == 0) { stamps[0] = t0; stamps[1] = t1; stamps[2] = bad; }
    for (int idx = lane; idx < 4096; idx += 64) { const int rr = idx >> 6, cc = idx & 63; Lout[idx] = S[rr * LS + cc]; }
    for (int idx = lane; idx < 64; idx += 64) Lout[4096 + idx] = Dr[idx];
}
// issue-rate probes: 16 independent accumulators, N rounds
template <int V>
__global__ void __launch_bounds__(64) rate_kernel(double* out, long long* stamps, int rounds) {
    double acc[16], b = out[threadIdx.x], c = out[64 + threadIdx.x];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    const long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (V == 0) acc[i] = __builtin_fma(b, c, acc[i]);
            if (V == 1) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(b), "v"(c));
            if (V == 2) acc[i] = __builtin_fma(bcast(b, i), c, acc[i]);
            if (V == 3) acc[i] = __builtin_fma(rowb<5>(acc[(i + 8) & 15]), c, acc[i]);
            if (V == 4) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(b), "v"(c));
        }
    }
    const long long t1 = wall_clock64();
    double s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[128 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[0] = t0; stamps[1] = t1; }
}
// dependent chain probes: latency of fma -> fma, rsq, mov_dpp -> fma
template <int V>
__global__ void __launch_bounds__(64) lat_kernel(double* out, long long* stamps, int rounds) {
    double x = out[threadIdx.x] + 1.5, c = out[64 + threadIdx.x] * 1e-9 + 1.0;
    const long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (V == 0) x = __builtin_fma(x, c, 1e-3);
            if (V == 1) x = __builtin_amdgcn_rsq(x) + 1.0;
            if (V == 2) x = __builtin_fma(rowb<3>(x), c, 1e-3);
            if (V == 3) x = rsqrt_newton(x) + 1.0;
            if (V == 4) x = __builtin_fma(bcast(x, 3), c, 1e-3);
        }
    }
    const long long t1 = wall_clock64();
    out[128 + threadIdx.x] = x;
    if (threadIdx.x == 0) { stamps[0] = t0; stamps[1] = t1; }
}
template <int V> double run_panel(const double* A, double* dL, long long* st, std::vector<double>& L) {
    long long s[3];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(panel_kernel<V>, dim3(1), dim3(64), 0, 0, A, dL, st, 200); hipDeviceSynchronize(); }
    hipMemcpy(s, st, 24, hipMemcpyDeviceToHost);
    hipMemcpy(L.data(), dL, (4096 + 64) * 8, hipMemcpyDeviceToHost);
    return (s[1] - s[0]) / 100.0 / 200;
}
int main() {
    const int n = 64;
    std::vector<double> h(n * n), L0(4160), L1(4160), L2(4160), L3(4160);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) h[i * n + j] = (i == j ? 70.0 : 0.0) + 1.0 / (1 + abs(i - j));
    double *A, *dL; long long* st;
    hipMalloc(&A, n * n * 8); hipMalloc(&dL, 4160 * 8); hipMalloc(&st, 64);
    hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice);
    const double tc = run_panel<3>(A, dL, st, L3);
    const double t0 = run_panel<0>(A, dL, st, L0);
    const double t1 = run_panel<1>(A, dL, st, L1);
    const double t2 = run_panel<2>(A, dL, st, L2);
    printf("copy only            %7.2f us\n4 panels readlane    %7.2f us\n4 panels mov_dpp+fma %7.2f us\n4 panels fmac_dpp    %7.2f us\n", tc, t0, t1, t2);
    double e1 = 0, e2 = 0, ed = 0;
    for (int i = 0; i < 4096; ++i) { e1 = fmax(e1, fabs(L1[i] - L0[i])); e2 = fmax(e2, fabs(L2[i] - L0[i])); }
    for (int i = 4096; i < 4160; ++i) ed = fmax(ed, fmax(fabs(L1[i] - L0[i]) / fabs(L0[i]), fabs(L2[i] - L0[i]) / fabs(L0[i])));
    printf("max |dpp1 - readlane| = %.2e   max |dpp2 - readlane| = %.2e   rel diff of 1/L_cc = %.2e\n", e1, e2, ed);
    double* out; hipMalloc(&out, 192 * 8); hipMemset(out, 0, 192 * 8);
    long long s[2];
    const char* rn[] = {"v_fma_f64", "v_fmac_f64_dpp", "2 readlane + fma", "mov_dpp(acc) + fma", "s_nop 1 + fmac_dpp"};
    const int rounds = 2000;
#define RATE(V) { for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(rate_kernel<V>, dim3(1), dim3(64), 0, 0, out, st, rounds); hipDeviceSynchronize(); } \
      hipMemcpy(s, st, 16, hipMemcpyDeviceToHost); printf("issue  %-22s %6.2f ns per op\n", rn[V], (s[1] - s[0]) * 10.0 / (rounds * 16)); }
    RATE(0) RATE(1) RATE(2) RATE(3) RATE(4)
    const char* ln[] = {"fma -> fma", "rsq + add", "mov_dpp -> fma", "rsqrt_newton + add", "2 readlane -> fma"};
#define LAT(V) { for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(lat_kernel<V>, dim3(1), dim3(64), 0, 0, out, st, rounds); hipDeviceSynchronize(); } \
      hipMemcpy(s, st, 16, hipMemcpyDeviceToHost); printf("chain  %-22s %6.2f ns per link\n", ln[V], (s[1] - s[0]) * 10.0 / (rounds * 16)); }
    LAT(0) LAT(1) LAT(2) LAT(3) LAT(4)
    return 0;
}
