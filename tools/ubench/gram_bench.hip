// What bounds gram_kernel at C2 (N = 4096, d = 6)?  The product kernel next to variants of the same loop:
//   V=1 no exp (v = dist), V=2 no stores (one conditional store per thread), V=3 stores only (v = constant),
//   V=4 product loop but a whole 64 x 64 tile per workgroup (16 entries per thread)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../gp_mpc_amd/csrc/gp_kernels.hpp"
using namespace gpmpc;

template <int D, int V, int ROWS>
__global__ void __launch_bounds__(256) gram_variant(const double* __restrict__ XT, const double* __restrict__ hyper,
                                                    double* __restrict__ K, int N, int Np) {
#pragma clang fp contract(off)
    constexpr int Q = 64 / ROWS;
    const int tn = blockIdx.x, tm = (int)blockIdx.y / Q, rq = (int)blockIdx.y % Q;
    if (tn > tm) return;
    __shared__ double X2r[D][ROWS], Qr[D][ROWS];
    const int tid = threadIdx.x, m0 = tm * 64 + ROWS * rq, n0 = tn * 64, c = tid & 63;
    const double* hy = hyper;
    for (int idx = tid; idx < ROWS * D; idx += 256) {
        const int dd = idx / ROWS, i = idx % ROWS;
        const double xr = XT[(long)dd * Np + m0 + i];
        X2r[dd][i] = 2.0 * xr;
        Qr[dd][i] = xr * xr;
    }
    double xc[D], qc[D], e2[D], ie2[D];
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
        xc[dd] = XT[(long)dd * Np + n0 + c];
        qc[dd] = xc[dd] * xc[dd];
        e2[dd] = hy[dd] * hy[dd];
        ie2[dd] = 1.0 / e2[dd];
    }
    const double sf2 = hy[D] * hy[D];
    __syncthreads();
    const int j = n0 + c;
    double keep = 0.0;
#pragma unroll 4
    for (int s = 0; s < ROWS / 4; ++s) {
        const int r = (tid >> 6) + 4 * s, i = m0 + r;
        double v;
        if (V == 3) {
            v = 1.5;
        } else if (j > i) {
            v = 0.0;
        } else {
            double dist = 0.0;
#pragma unroll
            for (int dd = 0; dd < D; ++dd) {
                const double t = (Qr[dd][r] + qc[dd]) - X2r[dd][r] * xc[dd];
                dist = div_by_const(t, e2[dd], ie2[dd], true) + dist;
            }
            v = V == 1 ? dist : sf2 * exp_lean(-0.5 * dist);
        }
        if (V == 2) keep += v;
        else K[(long)i * Np + j] = v;
    }
    if (V == 2 && keep == 123.456) K[(long)m0 * Np + j] = keep;
}


// rows through scalar loads: the row index of a wave is uniform, x_i and x_i^2 sit in SGPRs, no LDS, no barrier
template <int D, int ROWS, bool LEAN, int REP = 1>
__global__ void __launch_bounds__(256) gram_scalar(const double* __restrict__ XT, const double* __restrict__ hyper,
                                                   double* __restrict__ K, int N, int Np) {
#pragma clang fp contract(off)
    constexpr int Q = 64 / ROWS;
    const int tn = blockIdx.x, tm = (int)blockIdx.y / Q, rq = (int)blockIdx.y % Q;
    if (tn > tm) return;
    const int tid = threadIdx.x, m0 = tm * 64 + ROWS * rq, n0 = tn * 64, c = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* hy = hyper;
    double xc[D], qc[D], e2[D], ie2[D];
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
        xc[dd] = XT[(long)dd * Np + n0 + c];
        qc[dd] = xc[dd] * xc[dd];
        e2[dd] = hy[dd] * hy[dd];
        ie2[dd] = 1.0 / e2[dd];
    }
    const double sf2 = hy[D] * hy[D];
    const int j = n0 + c;
    for (int rep = 0; rep < REP; ++rep)
#pragma unroll 4
    for (int s = 0; s < ROWS / 4; ++s) {
        const int r = wv + 4 * s, i = m0 + r;
        double v;
        if (j > i) {
            v = 0.0;
        } else {
            double dist = 0.0;
#pragma unroll
            for (int dd = 0; dd < D; ++dd) {
                const double xr = XT[(long)dd * Np + i];
                const double t = (xr * xr + qc[dd]) - (2.0 * xr) * xc[dd];
                dist = div_by_const(t, e2[dd], ie2[dd], true) + dist;
            }
            v = LEAN ? sf2 * exp_lean(-0.5 * dist) : sf2 * exp(-0.5 * dist);
        }
        K[(long)i * Np + j] = v;
    }
}
template <int ROWS, bool LEAN, int REP = 1>
static float run_scalar(const double* XT, const double* hy, double* K, int N, int Np) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const dim3 grid(Np / 64, (Np / 64) * (64 / ROWS), 1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((gram_scalar<6, ROWS, LEAN, REP>), grid, dim3(256), 0, 0, XT, hy, K, N, Np);
    hipEventRecord(a);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((gram_scalar<6, ROWS, LEAN, REP>), grid, dim3(256), 0, 0, XT, hy, K, N, Np);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 20 * 1e3f;
}

template <int V, int ROWS>
static float run(const double* XT, const double* hy, double* K, int N, int Np) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const dim3 grid(Np / 64, (Np / 64) * (64 / ROWS), 1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((gram_variant<6, V, ROWS>), grid, dim3(256), 0, 0, XT, hy, K, N, Np);
    hipEventRecord(a);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((gram_variant<6, V, ROWS>), grid, dim3(256), 0, 0, XT, hy, K, N, Np);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 20 * 1e3f;
}

int main() {
    const int N = 4096, Np = 4096, d = 6;
    std::vector<double> x((size_t)d * Np), hy = {2, 2, 2, 2, 2, 2, 1.0, 1e-2};
    for (size_t i = 0; i < x.size(); ++i) x[i] = ((i * 2654435761u) % 10007) / 5000.0 - 1.0;
    double *XT, *H, *K, *jit;
    hipMalloc(&XT, x.size() * 8); hipMalloc(&H, 64); hipMalloc(&K, (size_t)Np * Np * 8); hipMalloc(&jit, 8);
    hipMemcpy(XT, x.data(), x.size() * 8, hipMemcpyHostToDevice); hipMemcpy(H, hy.data(), 64, hipMemcpyHostToDevice); hipMemset(jit, 0, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) launch_gram(0, dim3(Np / 64, Np / 64, 1), d, XT, H, jit, K, N, Np);
    hipEventRecord(a);
    for (int w = 0; w < 20; ++w) launch_gram(0, dim3(Np / 64, Np / 64, 1), d, XT, H, jit, K, N, Np);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("product gram_kernel<6>            %7.1f us\n", ms / 20 * 1e3);
    printf("variant full, 16 rows             %7.1f us\n", run<0, 16>(XT, H, K, N, Np));
    printf("variant full, 64 rows             %7.1f us\n", run<0, 64>(XT, H, K, N, Np));
    printf("variant no exp, 16 rows           %7.1f us\n", run<1, 16>(XT, H, K, N, Np));
    printf("variant no stores, 16 rows        %7.1f us\n", run<2, 16>(XT, H, K, N, Np));
    printf("variant no stores, 64 rows        %7.1f us\n", run<2, 64>(XT, H, K, N, Np));
    printf("variant stores only, 16 rows      %7.1f us\n", run<3, 16>(XT, H, K, N, Np));
    printf("variant stores only, 64 rows      %7.1f us\n", run<3, 64>(XT, H, K, N, Np));
    printf("scalar-load rows, 16 rows         %7.1f us\n", run_scalar<16, true>(XT, H, K, N, Np));
    printf("scalar-load rows, 32 rows         %7.1f us\n", run_scalar<32, true>(XT, H, K, N, Np));
    printf("scalar-load rows, 64 rows         %7.1f us\n", run_scalar<64, true>(XT, H, K, N, Np));
    printf("scalar-load rows, 64 rows, x4 work %6.1f us\n", run_scalar<64, true, 4>(XT, H, K, N, Np));
    printf("scalar-load rows, 64 rows, x8 work %6.1f us\n", run_scalar<64, true, 8>(XT, H, K, N, Np));
    printf("scalar-load rows, 64 rows, libexp %7.1f us\n", run_scalar<64, false>(XT, H, K, N, Np));
    hipMemsetAsync(K, 0, (size_t)Np * Np * 8, 0);
    hipEventRecord(a);
    for (int w = 0; w < 20; ++w) hipMemsetAsync(K, 0, (size_t)Np * Np * 8 / 2, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("hipMemsetAsync of 67 MB           %7.1f us\n", ms / 20 * 1e3);
    return 0;
}
