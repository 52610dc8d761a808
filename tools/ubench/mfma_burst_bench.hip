// Is sustained fp64 MFMA power-limited on MI355X?  (DESIGN.md section 10 inferred "~1.3 of 2.4 GHz for launches beyond
// ~10 ms" from one PMC quotient; VERDICT r02 #11 asked for a sweep.)  A pure v_mfma_f64_16x16x4_f64 loop -- 8 independent
// accumulators, GEMM-like operand pattern, one wave per SIMD on every CU, the form that reaches 76-78 TFLOP/s in
// mfma_issue_bench -- is launched with iteration counts that give bursts of ~0.1 ... ~60 ms.  Per burst length:
//   * the rate over the whole launch (HIP events), and the clock it implies: TFLOP/s / (CUs x 4 SIMD x 2048 flop / 64 cycles);
//   * the clock seen from INSIDE: every wave reads s_memtime (shader clock) and s_memrealtime (100 MHz wall clock) at its
//     start and end; and the RATE of the first and of the last eighth of the iterations from the wall clock alone
//     (does it sag during the burst?).  (s_memtime turned out to tick at a constant 2.39 GHz whatever the shader clock.)
// Also: back-to-back short bursts (is the limit a running average across launches?) and a cool-down series.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

// (4 blocks per CU allowed = at most 128 VGPRs: with more room hipcc keeps the accumulators in AGPRs and copies them
//  in and out every iteration, 142 instead of 64 cycles per MFMA -- the flaw of this repository's first MFMA benchmark)
// V = 0: operands in registers; V = 1: the four A operands of the next iteration are read from LDS (the fragment traffic of
// a GEMM inner loop); V = 2: as 1, and every wave streams 64 B per lane and iteration from a large HBM buffer (the
// operand stream of a memory-heavy product; the value is folded into an operand so that the load cannot be dropped).
template <int V>
__global__ void __launch_bounds__(256, 4) burst(double* __restrict__ out, const double* __restrict__ in, long iters, long long* stamps,
                                                const double* __restrict__ big, long big_elems) {
    const int t = threadIdx.x;
    __shared__ double lds[2048];
    for (int i = t; i < 2048; i += 256) lds[i] = in[i & 1023];
    __syncthreads();
    long gpos = ((long)blockIdx.x * 256 + t) * 8;
    double a0 = in[t], a1 = in[t + 256], a2 = in[t + 512], a3 = in[t + 768], b0 = in[t + 1024], b1 = in[t + 1280];
    d4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = d4{0, 0, 0, 0};
    const long e8 = iters / 8 > 0 ? iters / 8 : 1;
    long long s0 = __builtin_amdgcn_s_memtime(), r0 = wall_clock64(), s1 = 0, r1 = 0, s2 = 0, r2 = 0;
#define BODY c[0] = MFMA(a0, b0, c[0]); c[1] = MFMA(a0, b1, c[1]); c[2] = MFMA(a1, b0, c[2]); c[3] = MFMA(a1, b1, c[3]); \
             c[4] = MFMA(a2, b0, c[4]); c[5] = MFMA(a2, b1, c[5]); c[6] = MFMA(a3, b0, c[6]); c[7] = MFMA(a3, b1, c[7]); \
             if (V >= 1) { const double2 n0 = *reinterpret_cast<const double2*>(&lds[(2 * t + 2 * i) & 2046]);            \
                           const double2 n1 = *reinterpret_cast<const double2*>(&lds[(2 * t + 2 * i + 512) & 2046]);      \
                           a0 = n0.x; a1 = n0.y; a2 = n1.x; a3 = n1.y; }                                                  \
             if (V == 2) { const double4 g0 = *reinterpret_cast<const double4*>(&big[gpos]);                               \
                           const double4 g1 = *reinterpret_cast<const double4*>(&big[gpos + 4]);                           \
                           gpos += (long)gridDim.x * 256 * 8; if (gpos + 8 > big_elems) gpos = ((long)blockIdx.x * 256 + t) * 8; \
                           b0 += (g0.x + g0.w + g1.y) * 1e-300; }
    for (int i = 0; i < (int)e8; ++i) { BODY }
    s1 = __builtin_amdgcn_s_memtime(); r1 = wall_clock64();
    for (int i = 0; i < (int)(iters - 2 * e8); ++i) { BODY }
    s2 = __builtin_amdgcn_s_memtime(); r2 = wall_clock64();
    for (int i = 0; i < (int)e8; ++i) { BODY }
    long long s3 = __builtin_amdgcn_s_memtime(), r3 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
    out[(long)blockIdx.x * 256 + t] = s;
    if (t == 0 && stamps) {
        long long* p = stamps + (long)blockIdx.x * 8;
        p[0] = s0; p[1] = r0; p[2] = s1; p[3] = r1; p[4] = s2; p[5] = r2; p[6] = s3; p[7] = r3;
    }
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double *in, *out; long long* st;
    hipMalloc(&in, 2048 * 8); hipMalloc(&out, (size_t)cus * 256 * 8); hipMalloc(&st, (size_t)cus * 8 * 8);
    std::vector<double> h(2048, 1e-3); hipMemcpy(in, h.data(), 2048 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop_per_iter = 8.0 * 2048 * 4 * cus;            // 8 MFMA x 2048 flop x 4 waves (one per SIMD) x CUs
    const double peak_per_ghz = cus * 4 * 2048.0 / 64.0 * 1e9 * 1e-12;   // TFLOP/s per GHz of shader clock
    const long big_elems = (long)1 << 28;                          // 2 GB: far beyond the 256 MB Infinity Cache
    double* big; hipMalloc(&big, big_elems * 8); hipMemset(big, 0, big_elems * 8);
    int variant = 0;
    auto run = [&](long iters, bool print, const char* tag) {
        hipEventRecord(e0, 0);
        if (variant == 0) hipLaunchKernelGGL(burst<0>, dim3(cus), dim3(256), 0, 0, out, (const double*)in, iters, st, (const double*)big, big_elems);
        if (variant == 1) hipLaunchKernelGGL(burst<1>, dim3(cus), dim3(256), 0, 0, out, (const double*)in, iters, st, (const double*)big, big_elems);
        if (variant == 2) hipLaunchKernelGGL(burst<2>, dim3(cus), dim3(256), 0, 0, out, (const double*)in, iters, st, (const double*)big, big_elems);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> s((size_t)cus * 8);
        hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
        double f_all = 0, t_first = 0, t_last = 0;
        const long e8 = iters / 8 > 0 ? iters / 8 : 1;
        for (int w = 0; w < cus; ++w) {
            const long long* p = &s[(size_t)w * 8];
            f_all += (double)(p[6] - p[0]) / ((p[7] - p[1]) * 10.0);       // s_memtime ticks per ns
            t_first += (p[3] - p[1]) * 10.0;                               // ns of the first eighth (s_memrealtime: 100 MHz)
            t_last += (p[7] - p[5]) * 10.0;
        }
        const double tf = flop_per_iter * iters / (ms * 1e-3) * 1e-12;
        const double tf_first = flop_per_iter * e8 / (t_first / cus * 1e-9) * 1e-12, tf_last = flop_per_iter * e8 / (t_last / cus * 1e-9) * 1e-12;
        if (print)
            printf("%-14s iters %9ld  burst %8.3f ms  %6.2f TFLOP/s  implied clock %.3f GHz | first eighth %6.2f TFLOP/s (%.3f GHz)  last eighth %6.2f TFLOP/s (%.3f GHz)  [s_memtime %.3f ticks/ns]\n",
                   tag, iters, ms, tf, tf / peak_per_ghz, tf_first, tf_first / peak_per_ghz, tf_last, tf_last / peak_per_ghz, f_all / cus);
        return (double)ms;
    };
    printf("device %s, %d CUs; pure fp64 MFMA loop, one wave per SIMD; peak at 2.4 GHz = %.1f TFLOP/s\n", prop.name, cus, 2.4 * peak_per_ghz);
    run(20000, false, "warm-up");
    const long its[] = {2000, 5000, 12000, 25000, 60000, 120000, 250000, 500000, 1000000, 2000000, 4000000};
    for (long n : its) run(n, true, "single burst");
    const char* vn[] = {"", "operands re-read from LDS every iteration", "LDS operands + 64 B per lane and iteration streamed from HBM"};
    for (variant = 1; variant <= 2; ++variant) {
        printf("-- variant %d: %s\n", variant, vn[variant]);
        run(20000, false, "warm-up");
        const long its2[] = {5000, 25000, 120000, 500000, 2000000};
        for (long n : its2) run(n, true, variant == 1 ? "lds burst" : "lds+hbm burst");
    }
    variant = 0;
    printf("-- back to back: 40 equal bursts (does a running average across launches limit them?)\n");
    for (int i = 0; i < 40; ++i) { const bool p = i < 3 || i % 8 == 7; run(100000, p, "back-to-back"); }
    printf("-- after 200 ms of idling\n");
    hipDeviceSynchronize(); struct timespec ts = {0, 200000000}; nanosleep(&ts, nullptr);
    run(100000, true, "after idle");
    run(4000000, true, "long again");
    return 0;
}
