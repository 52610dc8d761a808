// quick self-test of the emulator: block reduce with barriers + shuffles + one MFMA
#include <hip/hip_runtime.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_reduce(const double* in, double* out, int n) {
    __shared__ double part[4];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += in[i];
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ void k_mfma(const double* A, const double* B, double* D) {  // A 16x4, B 4x16 row-major
    int l = threadIdx.x;
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
int main() {
    double in_[1000], out_[3]; double *in = in_, *out = out_;
    for (int i = 0; i < 1000; ++i) in[i] = i;
    hipLaunchKernelGGL(k_reduce, dim3(3), dim3(256), 0, 0, in, out, 1000); hipDeviceSynchronize();
    for (int b = 0; b < 3; ++b) if (out[b] != 499500.0) { printf("reduce FAIL %g\n", out[b]); return 1; }
    double A_[64], B_[64], D_[256]; double *A = A_, *B = B_, *D = D_;
    for (int i = 0; i < 64; ++i) { A[i] = i + 1; B[i] = 2 * i - 7; }
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, A, B, D); hipDeviceSynchronize();
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j];
        if (s != D[i * 16 + j]) { printf("mfma FAIL %d %d\n", i, j); return 1; }
    }
    hipDeviceSynchronize();
    printf("emu selftest OK\n");
    return 0;
}
