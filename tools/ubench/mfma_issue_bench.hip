// How must v_mfma_f64_16x16x4_f64 be issued to keep a SIMD's matrix pipe (64 cycles per instruction) busy?
// Variants of a pure MFMA loop, each at 1, 2 and 4 waves per SIMD, reported as cycles per MFMA per SIMD
// (64 = pipe saturated).  Motivation: the vendor DGEMM reaches ~95 % with 2 waves per SIMD, the first
// micro-benchmark of this repo (4 accumulators, same operands, back to back) only 62 %.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

template <int V>
__global__ void __launch_bounds__(256, 4) k(double* __restrict__ out, const double* __restrict__ in, int iters) {
    const int t = threadIdx.x;
    double a0 = in[t], a1 = in[t + 256], a2 = in[t + 512], a3 = in[t + 768], b0 = in[t + 1024], b1 = in[t + 1280];
    d4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = d4{0, 0, 0, 0};
    double f = a0;
    __shared__ double lds[1024];
    lds[t] = a0; lds[t + 256] = a1;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        if (V == 0) {          // 4 accumulators, same operands, back to back
            c[0] = MFMA(a0, b0, c[0]); c[1] = MFMA(a0, b0, c[1]); c[2] = MFMA(a0, b0, c[2]); c[3] = MFMA(a0, b0, c[3]);
            c[0] = MFMA(a0, b0, c[0]); c[1] = MFMA(a0, b0, c[1]); c[2] = MFMA(a0, b0, c[2]); c[3] = MFMA(a0, b0, c[3]);
        } else if (V == 1) {   // 8 accumulators, GEMM-like operand pattern (4 A x 2 B)
            c[0] = MFMA(a0, b0, c[0]); c[1] = MFMA(a0, b1, c[1]); c[2] = MFMA(a1, b0, c[2]); c[3] = MFMA(a1, b1, c[3]);
            c[4] = MFMA(a2, b0, c[4]); c[5] = MFMA(a2, b1, c[5]); c[6] = MFMA(a3, b0, c[6]); c[7] = MFMA(a3, b1, c[7]);
        } else if (V == 2) {   // as 1, one independent VALU op after every MFMA
#define STEP(j, aa, bb) c[j] = MFMA(aa, bb, c[j]); f = __builtin_fma(f, 1.0000001, 1e-9); __builtin_amdgcn_sched_barrier(0);
            STEP(0, a0, b0) STEP(1, a0, b1) STEP(2, a1, b0) STEP(3, a1, b1) STEP(4, a2, b0) STEP(5, a2, b1) STEP(6, a3, b0) STEP(7, a3, b1)
#undef STEP
        } else if (V == 3) {   // as 1, an LDS read feeding the NEXT iteration's operand after every second MFMA
            c[0] = MFMA(a0, b0, c[0]); c[1] = MFMA(a0, b1, c[1]); __builtin_amdgcn_sched_barrier(0);
            double n0 = lds[(t + i) & 1023]; __builtin_amdgcn_sched_barrier(0);
            c[2] = MFMA(a1, b0, c[2]); c[3] = MFMA(a1, b1, c[3]); __builtin_amdgcn_sched_barrier(0);
            double n1 = lds[(t + i + 64) & 1023]; __builtin_amdgcn_sched_barrier(0);
            c[4] = MFMA(a2, b0, c[4]); c[5] = MFMA(a2, b1, c[5]); __builtin_amdgcn_sched_barrier(0);
            double n2 = lds[(t + i + 128) & 1023]; __builtin_amdgcn_sched_barrier(0);
            c[6] = MFMA(a3, b0, c[6]); c[7] = MFMA(a3, b1, c[7]); __builtin_amdgcn_sched_barrier(0);
            double n3 = lds[(t + i + 192) & 1023]; __builtin_amdgcn_sched_barrier(0);
            a0 = n0; a1 = n1; a2 = n2; a3 = n3;
        } else if (V == 4) {   // 16 accumulators would not fit next to c[8]: 8 accumulators, s_nop 1 between MFMAs
#define STEP(j, aa, bb) c[j] = MFMA(aa, bb, c[j]); asm volatile("s_nop 1"); __builtin_amdgcn_sched_barrier(0);
            STEP(0, a0, b0) STEP(1, a0, b1) STEP(2, a1, b0) STEP(3, a1, b1) STEP(4, a2, b0) STEP(5, a2, b1) STEP(6, a3, b0) STEP(7, a3, b1)
#undef STEP
        } else if (V == 6) {   // 2 accumulators (the tile-owner worker's wave piece), back to back
            c[0] = MFMA(a0, b0, c[0]); c[1] = MFMA(a0, b1, c[1]); c[0] = MFMA(a1, b0, c[0]); c[1] = MFMA(a1, b1, c[1]);
            c[0] = MFMA(a2, b0, c[0]); c[1] = MFMA(a2, b1, c[1]); c[0] = MFMA(a3, b0, c[0]); c[1] = MFMA(a3, b1, c[1]);
        } else if (V == 7) {   // 1 accumulator
            c[0] = MFMA(a0, b0, c[0]); c[0] = MFMA(a0, b1, c[0]); c[0] = MFMA(a1, b0, c[0]); c[0] = MFMA(a1, b1, c[0]);
            c[0] = MFMA(a2, b0, c[0]); c[0] = MFMA(a2, b1, c[0]); c[0] = MFMA(a3, b0, c[0]); c[0] = MFMA(a3, b1, c[0]);
        } else if (V == 5) {   // as 1 with s_setprio 3 held for the whole loop
            if (i == 0) __builtin_amdgcn_s_setprio(3);
            c[0] = MFMA(a0, b0, c[0]); c[1] = MFMA(a0, b1, c[1]); c[2] = MFMA(a1, b0, c[2]); c[3] = MFMA(a1, b1, c[3]);
            c[4] = MFMA(a2, b0, c[4]); c[5] = MFMA(a2, b1, c[5]); c[6] = MFMA(a3, b0, c[6]); c[7] = MFMA(a3, b1, c[7]);
        }
    }
    double s = f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
    out[(long)blockIdx.x * 256 + t] = s + a0 + a1 + a2 + a3;
}

template <int V>
void run(const char* name, double* out, const double* in, int cus) {
    const int iters = 4096;
    for (int wps = 1; wps <= 4; wps *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<V>, dim3(cus * wps), dim3(256), 0, 0, out, in, 64);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<V>, dim3(cus * wps), dim3(256), 0, 0, out, in, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mf = 8.0 * iters * wps;            // MFMAs per SIMD
        const double tf = 2048.0 * 8.0 * iters * 4.0 * cus * wps / (ms * 1e-3) * 1e-12;
        printf("  %-58s waves/SIMD %d : %7.3f ms  %6.2f TFLOP/s  %6.1f ns per MFMA per SIMD\n", name, wps, ms, tf, ms * 1e6 / mf);
    }
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    double *out, *in;
    hipMalloc(&out, (size_t)p.multiProcessorCount * 4 * 256 * 8); hipMalloc(&in, 2048 * 8);
    hipMemset(in, 0, 2048 * 8);
    printf("%s, %d CUs; 64 cycles = 26.7 ns at 2.4 GHz, 29.2 ns at 2.19 GHz\n", p.gcnArchName, p.multiProcessorCount);
    run<0>("V0 4 acc, same operands, back to back", out, in, p.multiProcessorCount);
    run<1>("V1 8 acc, 4 A x 2 B operands, back to back", out, in, p.multiProcessorCount);
    run<2>("V2 = V1 + one VALU fma after every MFMA", out, in, p.multiProcessorCount);
    run<3>("V3 = V1 + an LDS operand fetch after every 2nd MFMA", out, in, p.multiProcessorCount);
    run<4>("V4 = V1 + s_nop 1 after every MFMA", out, in, p.multiProcessorCount);
    run<5>("V5 = V1 at s_setprio 3", out, in, p.multiProcessorCount);
    run<6>("V6 2 acc, back to back", out, in, p.multiProcessorCount);
    run<7>("V7 1 acc, back to back", out, in, p.multiProcessorCount);
    return 0;
}
