// Can the fp64 matrix pipe and the fp64 VALU run concurrently at full rate?  Per CU: MW waves issue
// v_mfma_f64_16x16x4_f64 back to back, VW waves issue v_fma_f64 back to back (16 independent chains).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MW, int VW>
__global__ void __launch_bounds__(64 * (MW + VW)) k_mix(double* out, int iters, double seed) {
    const int wave = threadIdx.x >> 6;
    const double a = seed + 1e-3 * (threadIdx.x % 61), b = 1.0 / seed - 1e-3 * (threadIdx.x % 59);
    double s = 0;
    if (wave < MW) {
        d4 c0 = d4{0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        s = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        double c[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)      // 64 FMAs per iteration = same 4*2048/128 ... see flop accounting below
#pragma unroll
                for (int i = 0; i < 16; ++i) c[i] = __builtin_fma(c[i], a, b);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += c[i];
    }
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MW, int VW>
void run(int blocks_per_cu, int iters) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * blocks_per_cu, threads = 64 * (MW + VW);
    double* out; hipMalloc(&out, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix<MW, VW>), dim3(blocks), dim3(threads), 0, 0, out, 64, 1.37);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_mix<MW, VW>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.37);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)blocks * MW * iters * 4 * 2048.0, vf = (double)blocks * VW * iters * 64 * 128.0;
    printf("MFMA waves/CU %2d  VALU waves/CU %2d : %8.3f ms   MFMA %6.2f TF   VALU %6.2f TF   sum %6.2f TF\n", MW * blocks_per_cu, VW * blocks_per_cu,
           ms, mf / ms * 1e-9, vf / ms * 1e-9, (mf + vf) / ms * 1e-9);
    hipFree(out);
}
int main() {
    run<4, 0>(4, 20000);   // 16 MFMA waves per CU
    run<0, 4>(4, 20000);   // 16 VALU waves per CU
    run<4, 4>(2, 20000);   // 8 + 8
    run<4, 4>(4, 20000);   // 16 + 16
    run<8, 4>(2, 20000);   // 16 + 8
    run<8, 8>(2, 20000);   // 16 + 16 (bigger blocks)
    run<12, 4>(2, 20000);  // 24 + 8
    return 0;
}
