// Micro-benchmark: issue rate / latency of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950, with the
// effective shader clock (s_memtime vs s_memrealtime).  Build: hipcc --offload-arch=gfx950 -O3 mfma_f64_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(double* out, long long* clk, int iters, double seed) {
    const double a = seed + 1e-3 * (threadIdx.x % 61), b = 1.0 / seed - 1e-3 * (threadIdx.x % 59);
    d4 c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = d4{0, 0, 0, 0};
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int NACC>
__global__ void __launch_bounds__(256) k_fma(double* out, long long* clk, int iters, double seed) {
    const double a = seed + 1e-3 * (threadIdx.x % 61), b = 1e-9;
    double c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = i;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_fma(c[i], a, b);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += c[i];
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <class F>
void run(const char* name, F launch, int blocks, int threads, int iters, int nacc, double flop_per_inst_per_wave) {
    double* out; long long* clk;
    hipMalloc(&out, (size_t)blocks * threads * 8);
    hipMalloc(&clk, (size_t)blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(out, clk, 64);            // warm-up
    hipEventRecord(e0, 0);
    launch(out, clk, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0; for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; wall += h[2 * b + 1]; }
    cyc /= blocks; wall /= blocks;
    const double insts = (double)iters * nacc;                       // per wave
    const double waves = (double)blocks * threads / 64;
    const double tflops = insts * waves * flop_per_inst_per_wave / (ms * 1e-3) * 1e-12;
    printf("%-34s blocks=%4d thr=%3d acc=%2d : %8.3f ms  %7.2f TFLOP/s  s_memtime ticks/inst/wave=%7.1f  realtime ticks=%9.0f (=> %.1f ns/inst/wave)\n",
           name, blocks, threads, nacc, ms, tflops, cyc / insts, wall, ms * 1e6 / insts);
    hipFree(out); hipFree(clk);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d clock=%d kHz wallclock rate=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, 0);
    const int CU = p.multiProcessorCount;
    const double mf = 2.0 * 16 * 16 * 4, vf = 2.0 * 64;
#define RUN_MFMA(NA, BL, TH, IT) run("mfma_f64_16x16x4", [&](double* o, long long* c, int it) { hipLaunchKernelGGL(k_mfma<NA>, dim3(BL), dim3(TH), 0, 0, o, c, it, 1.37); }, BL, TH, IT, NA, mf)
#define RUN_FMA(NA, BL, TH, IT) run("v_fma_f64", [&](double* o, long long* c, int it) { hipLaunchKernelGGL(k_fma<NA>, dim3(BL), dim3(TH), 0, 0, o, c, it, 1.37); }, BL, TH, IT, NA, vf)
    RUN_MFMA(1, 1, 64, 20000);          // dependent-chain latency, idle chip
    RUN_MFMA(4, 1, 64, 20000);          // single wave issue rate
    RUN_MFMA(8, 1, 64, 20000);
    RUN_MFMA(4, CU, 256, 20000);        // 1 wave / SIMD, whole chip
    RUN_MFMA(8, CU, 256, 20000);
    RUN_MFMA(4, 2 * CU, 256, 20000);    // 2 waves / SIMD
    RUN_MFMA(4, 4 * CU, 256, 20000);    // 4 waves / SIMD
    RUN_MFMA(4, CU, 256, 200000);       // long run (clock settles)
    RUN_FMA(1, 1, 64, 200000);
    RUN_FMA(8, 1, 64, 200000);
    RUN_FMA(8, CU, 256, 200000);
    RUN_FMA(8, 4 * CU, 256, 200000);
    return 0;
}
