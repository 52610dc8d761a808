// SE-ARD kernel-matrix kernels and the small per-point assembly / reduction kernels of the GP
// hot path.  Layouts: XT[d][Np] (inputs transposed so that a wave reads 64 consecutive training
// points per dimension), Y/alpha/w [Ny][Np], K/L/invL [Ny][Np][Np] row-major, KsT[Ny][Bp][Np]
// (one contiguous k_a(X, z_j) vector per test point = the K-contiguous B operand of the variance
// GEMM).  Np = N rounded up to 64; padded rows/columns of K are the identity, padded entries of
// every vector are exact zeros, so padded quantities never contribute.
#pragma once
#include "mfma_f64.hpp"

namespace gpmpc {

constexpr int DMAX = 16;  // max GP input dimension d = Nx + Nu

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// exp(x) in fp64 without the library's special-case handling: n = rint(x log2 e), Cody-Waite reduction
// r = x - n ln 2, degree-12 Taylor polynomial on |r| <= 0.347 (truncation 1.7e-16), v_ldexp_f64.
// 18 VALU instructions; arguments here are <= O(10), underflow flushes to 0 through ldexp.
__device__ __forceinline__ double exp_lean(double x) {
    const double n = rint(x * 1.4426950408889634074);
    double r = fma(-n, 6.93147180369123816490e-01, x);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.0 / 479001600.0;
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// exp(x) with a table of 2^(j / 2048) (16 KB, copied to LDS by the workgroup: exp_tab_fill):  x = (2048 n + j) ln2 / 2048 + r,
// |r| <= ln2 / 4096 = 1.7e-4, exp(x) = 2^n T[j] (1 + r + r^2/2 + r^3/6) (truncation r^4 / 24 = 3.4e-17).  The integer
// 2048 n + j is read off the low word of x * 2048 / ln2 + 1.5 * 2^52 (no rint / cvt), the reduction is Cody-Waite with the
// constants of exp_lean scaled by 2^-11 (exact), 2^n goes in through v_ldexp_f64 (underflow flushes to 0).  Eleven VALU
// instructions and one LDS read against the eighteen of exp_lean (three of them conversion-class, issued at a fraction of the
// fma rate); error <= 1 ulp of T[j] rounding + 0.6 ulp.  For the exact-moment pair sums (em_kernels.hpp), whose VALU time is
// this function; the K build keeps exp_lean (its bits are what the fixtures were checked with).
constexpr int EXPT_LOG2 = 11, EXPT_N = 1 << EXPT_LOG2;
__device__ __forceinline__ void exp_tab_fill(double* __restrict__ T_lds, const double* __restrict__ T_glob, int tid, int nthreads) {
    for (int i = tid; i < EXPT_N; i += nthreads) T_lds[i] = T_glob[i];
}
// Domain (r06): the integer n = rint(x 2048 / ln 2) sits in the mantissa of t; its part above the table index -- the power of
// two -- is taken from BOTH words of t (one funnel shift, v_alignbit_b32, where r01-r05 shifted the low word alone and n
// wrapped beyond |x| = 7.3e5, a test point 1 200 length scales away: ldexp then returned inf or garbage instead of 0 and the
// exact-moment covariance came out NaN), so it is exact for |n| < 2^42, |x| < 1.5e9 (5e4 length scales).  CLAMP adds one
// v_max_f64 that makes every x <= -1e9, -inf included (sf = 0 through gpmpc_set_factors), an exact zero: +1.8 % on the C3 step
// when always on (profiles/r06_em_exp_clamp_ab.txt), so the pair-sum kernels instantiate their sweep twice and take the
// clamped one only when the operands' magnitudes (em_operands_kernel) do not rule such arguments out.
template <bool CLAMP = true>
__device__ __forceinline__ double exp_tab(double x, const double* __restrict__ T_lds) {
    const double magic = 6755399441055744.0;                    // 1.5 * 2^52
    if (CLAMP) x = fmax(x, -1.0e9);
    const double t = fma(x, 2954.6394437405970050, magic);      // 2048 / ln 2
    const double nf = t - magic;
    // ONE reduction constant (r05: one VALU instruction of twelve less): RN(ln2) / 2048 is off by < 2^-54 relative, so
    // r = x - nf c (exact product inside the fma, one rounding) differs from the exact remainder by x 2^-54 at most -- a
    // relative perturbation of the ARGUMENT below its own rounding error (x is a sum of rounded terms)
    const double r = fma(-nf, 6.93147180559945309417e-01 / 2048.0, x);
    const int ki = __double2loint(t);
    const double tj = T_lds[ki & (EXPT_N - 1)];
    double p = fma(r, 1.0 / 6.0, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int e2 = (int)(((unsigned)__double2hiint(t) << (32 - EXPT_LOG2)) | ((unsigned)ki >> EXPT_LOG2));   // bits 11 .. 42 of n
    return ldexp(tj * p, e2);
}

// exp(x) with a table that occupies the 64 LDS banks exactly once: 32 doubles 2^(j / 32) = 256 bytes (every 64th entry of the
// table above, so the same correctly rounded values).  A ds_read_b64 is served in two groups of 32 lanes and a double covers
// two of the 64 four-byte banks: two lanes of a group either read the same entry (broadcast) or entries in different bank
// pairs -- no conflict whatever the indices are (the 2048-entry table: 5 extra LDS cycles per wave read at C3,
// profiles/r04_pmc_em_tab1_*).  x = (32 n + j) ln2 / 32 + r, |r| <= ln2 / 64 = 1.09e-2, degree-6 Taylor remainder
// (truncation r^7 / 5040 <= 3.5e-18), one reduction constant RN(ln2) / 32.  Fourteen full-rate VALU instructions and one
// conflict-free LDS read; <= 1 ulp of T[j] + 0.6 ulp + the argument perturbation x 2^-54 of the reduction (see exp_tab).
// Measured in the exact-moment pair sums (r05, C3): 5 % slower than the 2048-entry table -- three more VALU instructions
// per entry cost more than that table's bank conflicts -- so the pair sums keep exp_tab; the cross-covariance kernel,
// whose workgroups are too short-lived to copy a 16 KB table each, uses this one.
constexpr int EXPT32_N = 32;
__device__ static const double EXPT32[EXPT32_N] = {              // 2^(j / 32), correctly rounded (50-digit arithmetic)
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0,
    0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0,
    0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0,
    0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0,
    0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0,
    0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0, 0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0};
__device__ __forceinline__ void exp_tab32_fill(double* __restrict__ T_lds, int tid) {
    if (tid < EXPT32_N) T_lds[tid] = EXPT32[tid];
}
__device__ __forceinline__ double exp_tab32(double x, const double* __restrict__ T_lds) {
    const double magic = 6755399441055744.0;                    // 1.5 * 2^52
    x = fmax(x, -1.0e6);                                         // (n = rint(x 32 / ln 2) must stay inside 32 bits; exp(-1e6) = 0)
    const double t = fma(x, 46.166241308446829036, magic);      // 32 / ln 2
    const double nf = t - magic;
    const double r = fma(-nf, 6.93147180559945309417e-01 / 32.0, x);     // (one constant: see exp_tab)
    const int ki = __double2loint(t);
    const double tj = T_lds[ki & (EXPT32_N - 1)];
    double p = fma(r, 1.0 / 720.0, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(tj * p, ki >> 5);
}
// exp(-y) for an accumulated y (a squared distance): the sign goes into the constants, no negation instruction
// (y is clamped at 800 -- exp(-800) = 0 in fp64 -- so that the integer taken from the low word of t cannot wrap: beyond
//  y ~ 4.65e7, i.e. |x - z| / ell ~ 1e4, it did, and ldexp returned inf or garbage instead of 0; y = +inf from
//  -log(sf^2) with sf = 0 becomes 0 as well instead of NaN.  Unreachable with trained hyper-parameters, reachable through
//  gpmpc_set_factors; ADVICE r05.  One v_min_f64 per entry.)
__device__ __forceinline__ double exp_tab32_neg(double y, const double* __restrict__ T_lds) {
    const double magic = 6755399441055744.0;
    y = fmin(y, 800.0);
    const double t = fma(y, -46.166241308446829036, magic);
    const double nf = t - magic;                                 // rint(-y 32 / ln 2)
    const double s = fma(nf, 6.93147180559945309417e-01 / 32.0, y);      // s = -r (one constant: see exp_tab)
    const int ki = __double2loint(t);
    const double tj = T_lds[ki & (EXPT32_N - 1)];
    double p = fma(s, 1.0 / 720.0, -1.0 / 120.0);                // Taylor in r = -s: alternating signs
    p = fma(p, s, 1.0 / 24.0);
    p = fma(p, s, -1.0 / 6.0);
    p = fma(p, s, 0.5);
    p = fma(p, s, -1.0);
    p = fma(p, s, 1.0);
    return ldexp(tj * p, ki >> 5);
}

// t / b, correctly rounded, from the correctly rounded reciprocal y = RN(1 / b) (computed once per workgroup and
// dimension): q0 = t y, r = t - q0 b (exact in an fma), q = q0 + r y -- Markstein's division step: three full-rate
// instructions instead of the ~12 of the IEEE division sequence, and the SAME bits (200 M random pairs incl. mantissas
// with long runs of ones against `/` on the host: 0 mismatches).  The one case the theorem excludes, an all-ones
// mantissa of b, takes the real division (`exact` false; uniform per workgroup).
__device__ __forceinline__ double div_by_const(double t, double b, double y, bool exact) {
    if (!exact) return t / b;
    const double q0 = t * y;
    return fma(fma(-q0, b, t), y, q0);
}
__device__ __forceinline__ bool recip_is_safe(double b) {
    // (all-ones mantissa <=> the next representable number is a power of two: frexp mantissa 1 - 2^-53)
    int e;
    return frexp(fabs(b), &e) != 0.99999999999999988898;
}

// a1 + a3: K_a = sf^2 exp(-1/2 dist) + (sn^2 [+ jitter]) I on the lower triangle, with dist
// accumulated EXACTLY as the reference's numeric K build does (calc_cov_matrix optimize.py:314-318 /
// GP.covSEard gp_class.py:346-349): per input dimension the expanded form
//     dist = ((x_i^2 + x_j^2) - 2 (x_i x_j)) / ell^2 + dist
// in that operation order and without FMA contraction, so K matches numpy's K to the last bit of
// exp().  (On the reference's saved models, cond(K) up to 7e10, the choice of form moves chol(K) by
// 1e-10 relative -- the parity bar -- so the K build follows the form the fixtures were made with;
// the predict-side ks uses the direct-difference form that build_gp evaluates.)
// grid (Np/64, 4 Np/64, batch), 256 threads: a workgroup writes 16 rows x 64 columns of a 64 x 64 tile (a whole tile per
// workgroup made 2080 workgroups at C2 for 1024 resident ones: a third round for the last 32); tiles above the diagonal
// exit at once.  4 N (N+1) bytes per output are written, but the kernel is bound by its fp64 VALU work (63 operations
// per entry at d = 6 in the reference's operation order; 15 us at C2 against 11 us of HBM time).
// D (the input dimension) is a template parameter: the column point's coordinates and the per-dimension constants live
// in registers, the row point's are wave-uniform LDS reads, the distance loop is unrolled.  2 x_i is formed once per
// row (exact), the division by ell^2 is Markstein's three-instruction form (same bits), exp is exp_lean (< 1 ulp like
// the library's, without its special-case handling: -inf would give NaN, which the factorisation reports).
template <int D>
__global__ void __launch_bounds__(256) gram_kernel(const double* __restrict__ XT, const double* __restrict__ hyper,
                                                   const double* __restrict__ jitter, double* __restrict__ K,
                                                   int N, int Np, int tm0, int* __restrict__ zero_a, int n_zero_a,
                                                   int* __restrict__ zero_b, int n_zero_b) {
#pragma clang fp contract(off)
    const int tn = blockIdx.x, tm = (int)(blockIdx.y >> 2) + tm0, rq = blockIdx.y & 3, a = blockIdx.z;   // tm0: first tile row (gpmpc_append)
    // The hand-off flags and status words of the factorisation that follows are cleared HERE, by the first workgroup
    // (one above the diagonal if there is one: it has nothing else to do) -- as hipMemsetAsync calls they were three
    // fill kernels, each behind a dependency gap: 30 us between the K build and the chain's start at C2.
    if (blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x == (gridDim.x > 1 ? 1u : 0u)) {
        for (int i = threadIdx.x; i < n_zero_a; i += 256) zero_a[i] = 0;
        for (int i = threadIdx.x; i < n_zero_b; i += 256) zero_b[i] = 0;
    }
    if (tn > tm) return;
    __shared__ double X2r[D][16], Qr[D][16], Cs[2][D];
    __shared__ int safe_s;
    const int tid = threadIdx.x, m0 = tm * 64 + 16 * rq, n0 = tn * 64, c = tid & 63;
    const double* hy = hyper + (long)a * (D + 2);
    if (tid < 16 * D) {
        const int dd = tid >> 4, i = tid & 15;
        const double xr = XT[(long)dd * Np + m0 + i];
        X2r[dd][i] = 2.0 * xr;
        Qr[dd][i] = xr * xr;
    }
    if (tid == 255) {           // per-dimension constants once per workgroup
        bool ok = true;
        for (int dd = 0; dd < D; ++dd) {
            const double e = hy[dd] * hy[dd];
            Cs[0][dd] = e;
            Cs[1][dd] = 1.0 / e;
            ok = ok && recip_is_safe(e);
        }
        safe_s = ok ? 1 : 0;
    }
    double xc[D], qc[D];
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
        xc[dd] = XT[(long)dd * Np + n0 + c];
        qc[dd] = xc[dd] * xc[dd];
    }
    const double sf2 = hy[D] * hy[D], sn2 = hy[D + 1] * hy[D + 1], jit = jitter[a];
    __syncthreads();
    double e2[D], ie2[D];
#pragma unroll
    for (int dd = 0; dd < D; ++dd) { e2[dd] = Cs[0][dd]; ie2[dd] = Cs[1][dd]; }
    const bool fast_div = safe_s != 0;
    double* __restrict__ Ka = K + (long)a * Np * Np;
    const int j = n0 + c;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int r = (tid >> 6) + 4 * s, i = m0 + r;
        double v;
        if (i >= N || j >= N) {
            v = (i == j) ? 1.0 : 0.0;
        } else if (j > i) {
            v = 0.0;
        } else {
            double dist = 0.0;
#pragma unroll
            for (int dd = 0; dd < D; ++dd) {
                const double t = (Qr[dd][r] + qc[dd]) - X2r[dd][r] * xc[dd];      // 2 (x_i x_j) == (2 x_i) x_j exactly
                dist = div_by_const(t, e2[dd], ie2[dd], fast_div) + dist;
            }
            v = sf2 * exp_lean(-0.5 * dist);
            if (i == j) v = (v + sn2) + jit;
        }
        Ka[(long)i * Np + j] = v;
    }
}

inline void launch_gram(hipStream_t st, dim3 grid, int d, const double* XT, const double* hyper, const double* jitter, double* K,
                        int N, int Np, int tm0 = 0, int* zero_a = nullptr, int n_zero_a = 0, int* zero_b = nullptr, int n_zero_b = 0) {
#define GPMPC_GK(DD) case DD: hipLaunchKernelGGL((gram_kernel<DD>), dim3(grid.x, 4 * grid.y, grid.z), dim3(256), 0, st, XT, hyper, jitter, K, N, Np, tm0, zero_a, n_zero_a, zero_b, n_zero_b); break;
    switch (d) {
        GPMPC_GK(1) GPMPC_GK(2) GPMPC_GK(3) GPMPC_GK(4) GPMPC_GK(5) GPMPC_GK(6) GPMPC_GK(7) GPMPC_GK(8)
        GPMPC_GK(9) GPMPC_GK(10) GPMPC_GK(11) GPMPC_GK(12) GPMPC_GK(13) GPMPC_GK(14) GPMPC_GK(15) GPMPC_GK(16)
        default: break;
    }
#undef GPMPC_GK
}

// a1, two-input form: GP.covSEard gp_class.py:314-350 for arbitrary X[n1 x d], Z[n2 x d]; same
// per-dimension expanded-form accumulation as gram_kernel.  One thread per entry.
__global__ void __launch_bounds__(256) kernel_matrix_kernel(const double* __restrict__ X, const double* __restrict__ Z,
                                                            const double* __restrict__ ell, double sf2,
                                                            double* __restrict__ out, int n1, int n2, int d) {
#pragma clang fp contract(off)
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)n1 * n2) return;
    const int i = (int)(e / n2), j = (int)(e % n2);
    double dist = 0.0;
    for (int dd = 0; dd < d; ++dd) {
        const double x = X[(long)i * d + dd], z = Z[(long)j * d + dd];
        const double t = (x * x + z * z) - 2.0 * (x * z);
        dist = t / (ell[dd] * ell[dd]) + dist;
    }
    out[e] = sf2 * exp(-0.5 * dist);
}

// a9 (first half): ks_a(X, z_j) for JT test points per workgroup, written as KsT[a][j][:], fused
// with mean_a(z_j) = ks^T alpha_a (gp_functions.py:114-120,135) and, if JAC, with the analytic mean
// Jacobian J[j][a][dd] = sum_i alpha_i ks_i (X_i,dd - z_dd) / ell_dd^2 (what CasADi's AD yields for
// mean_jac_z, gp_functions.py:146-147).  grid (Bp/JT, Ny), 256 threads.
// D (the GP input dimension) is a template parameter so that a training point's coordinates live in
// registers and the distance loop is fully unrolled; the JT test points are wave-uniform LDS reads.
template <int D, int JT, bool JAC>
__global__ void __launch_bounds__(256) crosscov_kernel(const double* __restrict__ XT, const double* __restrict__ hyper,
                                                       const double* __restrict__ alpha, const double* __restrict__ Z,
                                                       double* __restrict__ KsT, double* __restrict__ meanT,
                                                       double* __restrict__ J, int N, int Np, int B, int Bp, int Ny,
                                                       double* __restrict__ cpart) {
    // gridDim.z > 1 (small batches, the MPC's shooting nodes): the training points are cut into gridDim.z
    // chunks so that more than Bp/JT x Ny workgroups exist; the partial sums go to cpart[chunk][a][j][1 + D]
    // and crosscov_finish_kernel adds them in a fixed order.
    // gridDim.x < Bp/JT (the launch next to a fit's tail, api_predict.inl): a workgroup walks the blocks of test points
    // blockIdx.x, blockIdx.x + gridDim.x, ... so that the launch occupies a few waves per CU only.
    const int a = blockIdx.y, tid = threadIdx.x;
    const int nch = gridDim.z, clen = ((Np + nch - 1) / nch + 255) / 256 * 256;
    const int ibeg = blockIdx.z * clen, iend = min(Np, ibeg + clen);
    constexpr int NR = JAC ? JT * (D + 1) : JT;
    // r05 arithmetic (the kernel is bound by VALU issue: 41 -> 2 D + 17 instructions per entry at the C2 shape): coordinates
    // pre-scaled by s_d = 1 / (sqrt 2 ell_d) -- the training point's once per JT test points, the test points' when staged --
    // so that  -1/2 sum_d (x_d - z_d)^2 / ell_d^2 = -sum_d (xs_d - zs_d)^2  is a subtraction and an fma per dimension; the
    // accumulator starts at -log sf^2, so sf^2 exp(.) needs no multiplication; exp through the 32-entry table that meets
    // every LDS bank once (exp_tab32_neg); the masks of padded training / test points only where a block has any.
    __shared__ double Zs[JT][D], w[D], red[4][NR], Et[EXPT32_N];
    const double* hy = hyper + (long)a * (D + 2);
    if (tid < D) w[tid] = 0.70710678118654752440 / fabs(hy[tid]);
    exp_tab32_fill(Et, tid);
    __syncthreads();
    const double neg_log_sf2 = -log(hy[D] * hy[D]);
    // alpha == nullptr (JAC == false only): the cross-covariances alone; mean_dot_kernel forms the mean later, once alpha
    // exists (the first prediction behind a fit runs next to the fit's tail, api_predict.inl)
    const double* __restrict__ al = alpha ? alpha + (long)a * Np : nullptr;
    for (int j0 = blockIdx.x * JT; j0 < Bp; j0 += gridDim.x * JT) {
    if (tid < JT * D) {
        const int jj = tid / D, dd = tid % D;
        Zs[jj][dd] = (j0 + jj < B) ? Z[(long)(j0 + jj) * D + dd] * w[dd] : 0.0;
    }
    __syncthreads();
    const bool all_points = j0 + JT <= B;
    double macc[JT], jacc[JAC ? JT : 1][D];
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) macc[jj] = 0.0;
    if (JAC) {
#pragma unroll
        for (int jj = 0; jj < JT; ++jj)
#pragma unroll
            for (int dd = 0; dd < D; ++dd) jacc[jj][dd] = 0.0;
    }
    double* __restrict__ out = KsT + ((long)a * Bp + j0) * Np;
    for (int i = ibeg + tid; i < iend; i += 256) {
        double x[D];
#pragma unroll
        for (int dd = 0; dd < D; ++dd) x[dd] = XT[(long)dd * Np + i] * w[dd];
        const double ai = al ? al[i] : 0.0;
        const bool live = i < N;
        const bool masked = !all_points || (i - tid + 255 >= N);     // (uniform over the workgroup)
#pragma unroll
        for (int jj = 0; jj < JT; ++jj) {
            double dist = neg_log_sf2, df[D];
#pragma unroll
            for (int dd = 0; dd < D; ++dd) {
                df[dd] = x[dd] - Zs[jj][dd];
                dist = fma(df[dd], df[dd], dist);
            }
            double ks = exp_tab32_neg(dist, Et);
            if (masked) ks = (live && j0 + jj < B) ? ks : 0.0;
            out[(long)jj * Np + i] = ks;
            macc[jj] = fma(ks, ai, macc[jj]);            // (the same mean bits with and without the Jacobian)
            if (JAC) {
                const double ka = ks * ai;
#pragma unroll
                for (int dd = 0; dd < D; ++dd) jacc[jj][dd] = fma(ka, df[dd], jacc[jj][dd]);
            }
        }
    }
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) {
        const double s = wave_sum(macc[jj]);
        if ((tid & 63) == 0) red[tid >> 6][jj] = s;
        if (JAC) {
#pragma unroll
            for (int dd = 0; dd < D; ++dd) {
                const double t = wave_sum(jacc[jj][dd]);
                if ((tid & 63) == 0) red[tid >> 6][JT + jj * D + dd] = t;
            }
        }
    }
    __syncthreads();
    if (nch > 1) {
        if (tid < NR) {
            const double v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            const int jj = tid < JT ? tid : (tid - JT) / D, e = tid < JT ? 0 : 1 + (tid - JT) % D;
            cpart[(((long)blockIdx.z * Ny + a) * Bp + j0 + jj) * (D + 1) + e] = e ? v * (2.0 * w[e - 1]) : v;   // (x - z) / ell^2 = 2 s (xs - zs)
        }
    } else {
        if (tid < JT && al) meanT[(long)a * Bp + j0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        if (JAC && tid >= JT && tid < NR) {
            const int e = tid - JT, jj = e / D, dd = e % D;
            if (j0 + jj < B)
                J[((long)(j0 + jj) * Ny + a) * D + dd] = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) * (2.0 * w[dd]);
        }
    }
    __syncthreads();          // Zs / red are rewritten by the next block of test points
    }
}

// adds the chunk partials of crosscov_kernel: one thread per (output a, point j, entry e); e = 0 mean, e > 0 J
__global__ void __launch_bounds__(256) crosscov_finish_kernel(const double* __restrict__ cpart, double* __restrict__ meanT,
                                                              double* __restrict__ J, int nch, int Ny, int B, int Bp,
                                                              int D, int with_jac) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int ne = D + 1;
    if (gid >= (long)Ny * Bp * ne) return;
    const int e = (int)(gid % ne), j = (int)((gid / ne) % Bp), a = (int)(gid / ((long)ne * Bp));
    if (e > 0 && !with_jac) return;
    double s = 0.0;
    for (int c = 0; c < nch; ++c) s += cpart[(((long)c * Ny + a) * Bp + j) * ne + e];
    if (e == 0) meanT[(long)a * Bp + j] = s;
    else if (j < B) J[((long)j * Ny + a) * D + e - 1] = s;
}

// dst[b][r][0 .. cols) = src[b][r][0 .. cols) for a batch of row-major rectangles (cols even, 16-byte aligned rows);
// grid (ceil(cols / 2 / 256), rows, batch)
__global__ void __launch_bounds__(256) copy_rect_kernel(const double* __restrict__ src, long lds_, long sS,
                                                        double* __restrict__ dst, long ldd, long sD, int cols,
                                                        const int* __restrict__ zmap = nullptr) {
    const int c = 2 * ((int)blockIdx.x * 256 + (int)threadIdx.x);
    if (c >= cols) return;
    const long bz = zmap ? zmap[blockIdx.z] : (int)blockIdx.z;
    const double2 v = *reinterpret_cast<const double2*>(src + bz * sS + (long)blockIdx.y * lds_ + c);
    *reinterpret_cast<double2*>(dst + bz * sD + (long)blockIdx.y * ldd + c) = v;
}

// mean_a(z_j) = ks^T alpha_a from the stored cross-covariances, for a crosscov_kernel launch that ran without alpha: the
// same thread -> training point mapping and the same reduction order as the fused sum, so both routes give the same bits.
// grid (Bp / JT, Ny), 256 threads.
template <int JT>
__global__ void __launch_bounds__(256) mean_dot_kernel(const double* __restrict__ KsT, const double* __restrict__ alpha,
                                                       double* __restrict__ meanT, int Np, int Bp) {
    const int j0 = blockIdx.x * JT, a = blockIdx.y, tid = threadIdx.x;
    __shared__ double red[4][JT];
    const double* __restrict__ al = alpha + (long)a * Np;
    const double* __restrict__ ks = KsT + ((long)a * Bp + j0) * Np;
    double macc[JT];
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) macc[jj] = 0.0;
    for (int i = tid; i < Np; i += 256) {
        const double ai = al[i];
#pragma unroll
        for (int jj = 0; jj < JT; ++jj) macc[jj] += ks[(long)jj * Np + i] * ai;
    }
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) {
        const double s = wave_sum(macc[jj]);
        if ((tid & 63) == 0) red[tid >> 6][jj] = s;
    }
    __syncthreads();
    if (tid < JT) meanT[(long)a * Bp + j0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

constexpr int CROSSCOV_JT = 8;       // test points per workgroup (4 when the Jacobian is accumulated as well)
constexpr int CROSSCOV_SMALL_B = 64; // up to this many (padded) test points the training points are chunked ...
constexpr int CROSSCOV_CHUNKS = 8;   // ... into this many pieces (gridDim.z)
#ifdef GPMPC_EMULATED
constexpr int CROSSCOV_CHUNK_MIN_NP = 512;    // (small in the emulated build so that the CPU tests reach the path)
#else
constexpr int CROSSCOV_CHUNK_MIN_NP = 2048;
#endif

template <int D>
inline void launch_crosscov_d(hipStream_t st, const double* XT, const double* hyper, const double* alpha,
                              const double* Z, double* KsT, double* meanT, double* J, int N, int Np, int B, int Bp,
                              int Ny, double* cpart, int nch, int max_wgs) {
    if (!cpart) nch = 1;
    auto gx = [&](int blocks) { return max_wgs > 0 && max_wgs < blocks ? max_wgs : blocks; };   // (throttled: see the kernel)
    if (J)
        hipLaunchKernelGGL((crosscov_kernel<D, 4, true>), dim3(gx(Bp / 4), Ny, nch), dim3(256), 0, st, XT, hyper, alpha, Z, KsT,
                           meanT, J, N, Np, B, Bp, Ny, cpart);
    else
        hipLaunchKernelGGL((crosscov_kernel<D, CROSSCOV_JT, false>), dim3(gx(Bp / CROSSCOV_JT), Ny, nch), dim3(256), 0, st, XT,
                           hyper, alpha, Z, KsT, meanT, J, N, Np, B, Bp, Ny, cpart);
    if (nch > 1) {
        const long items = (long)Ny * Bp * (D + 1);
        hipLaunchKernelGGL(crosscov_finish_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, cpart, meanT, J, nch,
                           Ny, B, Bp, D, J ? 1 : 0);
    }
}

inline void launch_crosscov(hipStream_t st, int d, const double* XT, const double* hyper, const double* alpha,
                            const double* Z, double* KsT, double* meanT, double* J, int N, int Np, int B, int Bp,
                            int Ny, double* cpart = nullptr, int nch = 1, int max_wgs = 0) {
#define GPMPC_CC(DD) case DD: launch_crosscov_d<DD>(st, XT, hyper, alpha, Z, KsT, meanT, J, N, Np, B, Bp, Ny, cpart, nch, max_wgs); break;
    switch (d) {
        GPMPC_CC(1) GPMPC_CC(2) GPMPC_CC(3) GPMPC_CC(4) GPMPC_CC(5) GPMPC_CC(6) GPMPC_CC(7) GPMPC_CC(8)
        GPMPC_CC(9) GPMPC_CC(10) GPMPC_CC(11) GPMPC_CC(12) GPMPC_CC(13) GPMPC_CC(14) GPMPC_CC(15) GPMPC_CC(16)
        default: break;
    }
#undef GPMPC_CC
}

// a9 (second half): var_a(z_j) = sf_a^2 - sum over row tiles of the column sums of squares
// written by the variance kernels (gp_functions.py:125-126,136: kss = sf^2, no noise), and the
// transposition of mean to the caller's [B][Ny] layout.  One workgroup per test point, fixed-order
// (deterministic) reduction over the row tiles.  grid (B), 256 threads.
// partm (optional): the mean as per-row-tile partial sums of the variance product's fused reduction instead of meanT.
__global__ void __launch_bounds__(256) var_finish_kernel(const double* __restrict__ part, const double* __restrict__ meanT,
                                                         const double* __restrict__ hyper, double* __restrict__ mean,
                                                         double* __restrict__ var, int B, int Bp, int Ny, int d,
                                                         int tilesM, const double* __restrict__ partm = nullptr) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ double red[4];
    for (int a = 0; a < Ny; ++a) {
        if (mean && partm) {
            double s = 0.0;
            for (int t = tid; t < tilesM; t += 256) s += partm[((long)a * tilesM + t) * Bp + b];
            s = wave_sum(s);
            if ((tid & 63) == 0) red[tid >> 6] = s;
            __syncthreads();
            if (tid == 0) mean[(long)b * Ny + a] = (red[0] + red[1]) + (red[2] + red[3]);
            __syncthreads();
        }
        if (var) {
            double s = 0.0;
            for (int t = tid; t < tilesM; t += 256) s += part[((long)a * tilesM + t) * Bp + b];
            s = wave_sum(s);
            if ((tid & 63) == 0) red[tid >> 6] = s;
            __syncthreads();
            if (tid == 0) {
                const double sf = hyper[(long)a * (d + 2) + d];
                var[(long)b * Ny + a] = sf * sf - ((red[0] + red[1]) + (red[2] + red[3]));
            }
            __syncthreads();
        }
        if (mean && !partm && tid == 0) mean[(long)b * Ny + a] = meanT[(long)a * Bp + b];
    }
}

// Small-batch predictive variance (the per-shooting-node pattern of the MPC, B <= 8): stream the lower
// triangle of L^-1 ONCE from HBM and form v_i = sum_k invL[i][k] ks_j[k] for all NB test points at once,
// one wave per row (16 B per lane, 1 KB per wave-instruction), rows dealt so that every workgroup gets
// the same amount of the triangle.  HBM-read bound: 4 N (N+1) bytes per output for up to 8 predictions.
// grid (Np / (4 rpw), Ny), 256 threads: 4 waves x rpw rows; part[a][blockIdx.x][j] = sum over the block's rows of v_i^2.
// rpw (rows per wave, 1 .. 8; var_small_rows_per_wave): 8 when that still gives every CU four workgroups (C3: 1536), fewer
// for small models -- at N = 4096, Ny = 1 eight rows per wave are 128 workgroups, half a workgroup per CU and two waves
// per CU with one or two loads in flight each: 1.35 TB/s (r06 bench, secondary.b1); one row per wave: 1024 workgroups.
template <int NB>
__global__ void __launch_bounds__(256) var_small_kernel(const double* __restrict__ Inv, const double* __restrict__ KsT,
                                                        double* __restrict__ part, int Np, int Bp, int rpw) {
    const int a = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = gridDim.x;
    __shared__ double red[4][NB];
    const double* __restrict__ Ia = Inv + (long)a * Np * Np;
    const double* __restrict__ ks = KsT + (long)a * Bp * Np;
    double sq[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) sq[j] = 0.0;
    for (int rr = 0; rr < rpw; ++rr) {
        // rows are dealt block-cyclically over the workgroups: balanced triangular work
        const int i = ((rr * 4 + wave) * nblk + (int)blockIdx.x);
        const double* __restrict__ row = Ia + (long)i * Np;
        double acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = 0.0;
        for (int k = 2 * lane; k <= i; k += 128) {
            const double2 l = *reinterpret_cast<const double2*>(row + k);   // row[i+1] is an exact zero (upper part)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const double2 x = *reinterpret_cast<const double2*>(ks + (long)j * Np + k);
                acc[j] += l.x * x.x + l.y * x.y;
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const double v = wave_sum(acc[j]);
            sq[j] += v * v;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NB; ++j) red[wave][j] = sq[j];
    }
    __syncthreads();
    if (tid < NB) part[((long)a * nblk + blockIdx.x) * Bp + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// rows per wave of var_small_kernel: the most (<= 8) that still leaves four workgroups per CU, else one
inline int var_small_rows_per_wave(int Np, int Ny, int cus) {
    for (int rpw = 8; rpw > 1; rpw /= 2)
        if ((long)(Np / (4 * rpw)) * Ny >= 4L * cus && Np % (4 * rpw) == 0) return rpw;
    return 1;
}

// a10 build_TA_cov gp_functions.py:152-173: cov[b] = diag(var[b]) + J[b] Sigma[b] J[b]^T
// (Sigma == nullptr -> the 'ME' covariance diag(var), gp_functions.py:143).  One thread per entry.
__global__ void __launch_bounds__(256) cov_assemble_kernel(const double* __restrict__ var, const double* __restrict__ J,
                                                           const double* __restrict__ Sigma, double* __restrict__ cov,
                                                           int B, int Ny, int d) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)B * Ny * Ny) return;
    const int c = (int)(e % Ny), a = (int)((e / Ny) % Ny);
    const long b = e / ((long)Ny * Ny);
    double v = (a == c) ? var[b * Ny + a] : 0.0;
    if (Sigma) {
        const double* Ja = J + (b * Ny + a) * d;
        const double* Jc = J + (b * Ny + c) * d;
        const double* S = Sigma + b * d * d;
        for (int p = 0; p < d; ++p) {
            double t = 0.0;
            for (int q = 0; q < d; ++q) t += S[p * d + q] * Jc[q];
            v += Ja[p] * t;
        }
    }
    cov[e] = v;
}

// Second-order information of the predictor at a test point (what a casadi Callback for
// GP.__predict must hand to IPOPT; the reference gets it from CasADi's AD of gp_functions.py:114-147):
//   Hm[b][a][p][q] = d^2 mean_a / dz_p dz_q = sum_i alpha_i ks_i r_ip r_iq - delta_pq mean_a / l_p^2,
//   dvar[b][a][p]  = d var_a / dz_p        = -2 sum_i u_i ks_i r_ip,
// with r_ip = (X_ip - z_p) / l_ap^2 and u = K_a^-1 ks (UT, one GEMM for the batch).  HBM-read bound: the two
// N-vectors ks and u per (point, output).  D is a template parameter: the D (D + 1) / 2 + D running sums live
// in registers.  grid (B, Ny), 256 threads; fixed-order reduction (deterministic).
template <int D>
__global__ void __launch_bounds__(256) sens_kernel(const double* __restrict__ XT, const double* __restrict__ Z,
                                                   const double* __restrict__ hyper, const double* __restrict__ alpha,
                                                   const double* __restrict__ KsT, const double* __restrict__ UT,
                                                   double* __restrict__ Hm, double* __restrict__ dvar, int N, int Np,
                                                   int Bp, int Ny) {
    constexpr int NH = D * (D + 1) / 2, NS = NH + D + 1;
    const int b = blockIdx.x, a = blockIdx.y, tid = threadIdx.x;
    __shared__ double red[4][NS];
    __shared__ double zs[D], w[D];
    const double* hy = hyper + (long)a * (D + 2);
    if (tid < D) {
        zs[tid] = Z[(long)b * D + tid];
        w[tid] = 1.0 / (hy[tid] * hy[tid]);
    }
    __syncthreads();
    const double* __restrict__ ks = KsT + ((long)a * Bp + b) * Np;
    const double* __restrict__ u = UT + ((long)a * Bp + b) * Np;
    const double* __restrict__ al = alpha + (long)a * Np;
    double acc[NS];
#pragma unroll
    for (int e = 0; e < NS; ++e) acc[e] = 0.0;
    for (int i = tid; i < N; i += 256) {
        const double k = ks[i], ka = k * al[i], ku = k * u[i];
        double r[D];
#pragma unroll
        for (int p = 0; p < D; ++p) r[p] = (XT[(long)p * Np + i] - zs[p]) * w[p];
        int e = 0;
#pragma unroll
        for (int p = 0; p < D; ++p)
#pragma unroll
            for (int q = 0; q <= p; ++q, ++e) acc[e] += ka * r[p] * r[q];
#pragma unroll
        for (int p = 0; p < D; ++p) acc[NH + p] += ku * r[p];
        acc[NH + D] += ka;
    }
#pragma unroll
    for (int e = 0; e < NS; ++e) {
        const double t = wave_sum(acc[e]);
        if ((tid & 63) == 0) red[tid >> 6][e] = t;
    }
    __syncthreads();
    if (tid < NS) red[0][tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    __syncthreads();
    const double mean = red[0][NH + D];
    if (tid < D * D) {
        const int p = tid / D, q = tid % D, hi = p > q ? p : q, lo = p > q ? q : p;
        double v = red[0][hi * (hi + 1) / 2 + lo];
        if (p == q) v -= mean * w[p];
        Hm[(((long)b * Ny + a) * D + p) * D + q] = v;
    }
    if (tid < D) dvar[((long)b * Ny + a) * D + tid] = -2.0 * red[0][NH + tid];
}

inline void launch_sens(hipStream_t st, int d, const double* XT, const double* Z, const double* hyper, const double* alpha,
                        const double* KsT, const double* UT, double* Hm, double* dvar, int N, int Np, int B, int Bp, int Ny) {
#define GPMPC_SK(DD) case DD: hipLaunchKernelGGL((sens_kernel<DD>), dim3(B, Ny), dim3(256), 0, st, XT, Z, hyper, alpha, KsT, \
                                                 UT, Hm, dvar, N, Np, Bp, Ny); break;
    switch (d) {
        GPMPC_SK(1) GPMPC_SK(2) GPMPC_SK(3) GPMPC_SK(4) GPMPC_SK(5) GPMPC_SK(6) GPMPC_SK(7) GPMPC_SK(8)
        GPMPC_SK(9) GPMPC_SK(10) GPMPC_SK(11) GPMPC_SK(12) GPMPC_SK(13) GPMPC_SK(14) GPMPC_SK(15) GPMPC_SK(16)
        default: break;
    }
#undef GPMPC_SK
}

// ---- prior mean functions (gp_functions.py:25-69): m(x) = 0 | c | a^T x + c | a^T x^2 + b^T x + c ----------------
// Parameter row (what follows sn in a hyper row, gp_functions.py:52-62): const [c]; linear [a_1..a_d, c];
// polynomial [a_1..a_d, b_1..b_d, c].  MEAN_* mirror GPMPC_MEAN_* of include/gpmpc.h.
constexpr int MEAN_ZERO = 0, MEAN_CONST = 1, MEAN_LINEAR = 2, MEAN_POLY = 3;
constexpr int MPW = 2 * DMAX + 1;   // stride of a mean-parameter row on the device
__host__ __device__ inline int mean_param_count(int kind, int d) {
    return kind == MEAN_CONST ? 1 : kind == MEAN_LINEAR ? d + 1 : kind == MEAN_POLY ? 2 * d + 1 : 0;
}
// m(x) for a point whose coordinates are x[k * stride], k < d
__device__ __forceinline__ double mean_eval(int kind, const double* __restrict__ mp, const double* __restrict__ x,
                                            long stride, int d) {
    if (kind == MEAN_CONST) return mp[0];
    double s = 0.0;
    if (kind == MEAN_LINEAR) {
        for (int k = 0; k < d; ++k) s += mp[k] * x[k * stride];
        return s + mp[d];
    }
    for (int k = 0; k < d; ++k) {
        const double xv = x[k * stride];
        s += mp[k] * (xv * xv) + mp[d + k] * xv;
    }
    return s + mp[2 * d];
}

// Yc[a][i] = Y[a][i] - m_a(x_i): the vector alpha and the NLL are formed from (optimize.py:75,285,494).
// grid (Np/256, batch), 256 threads.  mpar: [batch][MPW].
__global__ void __launch_bounds__(256) mean_resid_kernel(const double* __restrict__ XT, const double* __restrict__ Y,
                                                         const double* __restrict__ mpar, double* __restrict__ Yc,
                                                         int kind, int N, int Np, int d, long sy) {
    const int i = blockIdx.x * 256 + threadIdx.x, a = blockIdx.y;
    if (i >= Np) return;
    double v = 0.0;
    if (i < N) v = Y[(long)a * sy + i] - mean_eval(kind, mpar + (long)a * MPW, XT + i, Np, d);
    Yc[(long)a * Np + i] = v;
}

// d NLL / d (mean parameters) = -alpha^T dm/dtheta (from NLL = 1/2 (y-m)^T K^-1 (y-m) + ...):
// const: -sum alpha; linear a_k: -sum alpha_i x_ik; polynomial a_k: -sum alpha_i x_ik^2, b_k: -sum alpha_i x_ik.
// One workgroup, one wave per parameter in turn, fixed order.  out[count].
__global__ void __launch_bounds__(256) mean_grad_kernel(const double* __restrict__ XT, const double* __restrict__ alpha,
                                                        double* __restrict__ out, int kind, int N, int Np, int d, int gstride = 0,
                                                        const int* __restrict__ zmap = nullptr) {
    const int count = mean_param_count(kind, d), lane = threadIdx.x & 63;
    const int b = zmap ? zmap[blockIdx.x] : (int)blockIdx.x;   // grid (batch): alpha number b, out + b * gstride (zmap: a subset)
    alpha += (long)b * Np;
    out += (long)b * gstride;
    for (int e = threadIdx.x >> 6; e < count; e += 4) {
        const bool is_c = e == count - 1;
        const int k = (kind == MEAN_POLY && e >= d) ? e - d : e;
        const bool sq = kind == MEAN_POLY && e < d;
        double s = 0.0;
        for (int i = lane; i < N; i += 64) {
            const double xv = is_c ? 1.0 : XT[(long)k * Np + i];
            s += alpha[i] * (sq ? xv * xv : xv);
        }
        s = wave_sum(s);
        if (lane == 0) out[e] = -s;
    }
}

// build_gp with meanFunc (gp_functions.py:131,135): mean_a(z) += m_a(z); its derivatives follow: J_a += a (+ 2 a z
// + b), Hm_a += diag(2 a).  One thread per (point, output).  Any of mean / J / Hm may be NULL.
__global__ void __launch_bounds__(256) mean_add_kernel(const double* __restrict__ Z, const double* __restrict__ mpar,
                                                       double* __restrict__ mean, double* __restrict__ J,
                                                       double* __restrict__ Hm, int kind, int B, int Ny, int d) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)B * Ny) return;
    const int a = (int)(e % Ny);
    const long b = e / Ny;
    const double* mp = mpar + (long)a * MPW;
    const double* z = Z + b * d;
    if (mean) mean[e] += mean_eval(kind, mp, z, 1, d);
    if (J && kind >= MEAN_LINEAR)
        for (int k = 0; k < d; ++k) J[e * d + k] += kind == MEAN_LINEAR ? mp[k] : 2.0 * mp[k] * z[k] + mp[d + k];
    if (Hm && kind == MEAN_POLY)
        for (int k = 0; k < d; ++k) Hm[(e * d + k) * d + k] += 2.0 * mp[k];
}

// a17 on the device: input of step t of an uncertainty-propagation roll-out from the output of step t-1
// (GP.predict_compare's loop gp_class.py:777-804 with GP.predict's re-standardisation :253-261 folded in):
//   z_t = [sa * mean_{t-1} + sb, u_t],   Sigma_t[:Ny,:Ny] = cov_{t-1}  (the other blocks keep their initial values).
// One workgroup of 64 threads.
__global__ void __launch_bounds__(64) rollout_feed_kernel(const double* __restrict__ mean_prev, const double* __restrict__ cov_prev,
                                                          const double* __restrict__ u_t, const double* __restrict__ sa,
                                                          const double* __restrict__ sb, double* __restrict__ z,
                                                          double* __restrict__ Sigma, int Ny, int d,
                                                          const double* __restrict__ Kz = nullptr,
                                                          const double* __restrict__ k0 = nullptr,
                                                          const double* __restrict__ Kc = nullptr,
                                                          double* __restrict__ u_out = nullptr) {
    // Kz != NULL: state feedback (gp_class.py:789-790,797-803): u_t = Kz mean_{t-1} + k0 (the gain with GP.predict's
    // standardisation folded in) and the input covariance [[C, C Kc^T], [Kc C, Kc C Kc^T]] with C = cov_{t-1}.
    const int tid = threadIdx.x, Nu = d - Ny;
    for (int e = tid; e < d; e += 64) {
        double v;
        if (e < Ny) v = sa[e] * mean_prev[e] + sb[e];
        else if (Kz) {
            v = k0[e - Ny];
            for (int c = 0; c < Ny; ++c) v += Kz[(e - Ny) * Ny + c] * mean_prev[c];
            if (u_out) u_out[e - Ny] = v;
        } else v = u_t[e - Ny];
        z[e] = v;
    }
    for (int e = tid; e < Ny * Ny; e += 64) Sigma[(e / Ny) * d + e % Ny] = cov_prev[e];
    if (Kc) {
        for (int e = tid; e < Ny * Nu; e += 64) {           // cov_xu = C Kc^T  [Ny x Nu]
            const int r = e / Nu, q = e % Nu;
            double v = 0.0;
            for (int c = 0; c < Ny; ++c) v += cov_prev[r * Ny + c] * Kc[q * Ny + c];
            Sigma[r * d + Ny + q] = v;
            Sigma[(Ny + q) * d + r] = v;
        }
        for (int e = tid; e < Nu * Nu; e += 64) {           // covar_u = Kc C Kc^T
            const int p = e / Nu, q = e % Nu;
            double v = 0.0;
            for (int r = 0; r < Ny; ++r) {
                double t = 0.0;
                for (int c = 0; c < Ny; ++c) t += cov_prev[r * Ny + c] * Kc[q * Ny + c];
                v += Kc[p * Ny + r] * t;
            }
            Sigma[(Ny + p) * d + Ny + q] = v;
        }
    }
}

// The open-loop hand-over of rollout_feed_kernel for M trajectories that advance in lock-step (gpmpc_rollout_multi): workgroup m
// turns (mean, cov) of trajectory m at step t - 1 and its control u_t into its next input (z, Sigma).  grid (M), 64 threads.
__global__ void __launch_bounds__(64) rollout_feed_multi_kernel(const double* __restrict__ mean_prev, const double* __restrict__ cov_prev,
                                                                const double* __restrict__ u_t, const double* __restrict__ sa,
                                                                const double* __restrict__ sb, double* __restrict__ z,
                                                                double* __restrict__ Sigma, int Ny, int d, int nu1) {
    const int m = blockIdx.x, tid = threadIdx.x;
    const double* mp = mean_prev + (long)m * Ny;
    const double* cp = cov_prev + (long)m * Ny * Ny;
    const double* up = u_t + (long)m * nu1;
    double* zm = z + (long)m * d;
    double* Sm = Sigma + (long)m * d * d;
    for (int e = tid; e < d; e += 64) zm[e] = e < Ny ? sa[e] * mp[e] + sb[e] : up[e - Ny];
    for (int e = tid; e < Ny * Ny; e += 64) Sm[(e / Ny) * d + e % Ny] = cp[e];
}

// Matrix-vector products with the explicit factors (a5: alpha = L^-T (L^-1 y), optimize.py:353-354,494;
// beta = K^-1 y, gp_functions.py:383).  HBM-read bound: 4 N^2 bytes for a triangular operand.
// out[i] = sum_{k < (lower ? i+1 : Np)} A[i][k] x[k]: one wave per row, lanes stride the row (512 B
// coalesced reads), butterfly reduction.  grid (Np/4, batch), 256 threads.
__global__ void __launch_bounds__(256) gemv_rows_kernel(const double* __restrict__ A, const double* __restrict__ x,
                                                        double* __restrict__ out, int Np, long sA, long sx, long so,
                                                        int lower) {
    // 16-byte loads, two independent 1 KB wave loads in flight per iteration (rows are 512-byte aligned: Np % 64 == 0)
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const double2* __restrict__ row = reinterpret_cast<const double2*>(A + (long)blockIdx.y * sA + (long)i * Np);
    const double2* __restrict__ xv = reinterpret_cast<const double2*>(x + (long)blockIdx.y * sx);
    const int kend = lower ? i + 1 : Np;                 // entries beyond the diagonal of a lower operand are exact zeros
    const int nq = (kend + 1) >> 1;                      // double2 chunks (the odd tail multiplies a stored zero)
    double s0 = 0.0, s1 = 0.0;
    int q = lane;
    for (; q + 64 < nq; q += 128) {
        const double2 a0 = row[q], a1 = row[q + 64], b0 = xv[q], b1 = xv[q + 64];
        s0 += a0.x * b0.x + a0.y * b0.y;
        s1 += a1.x * b1.x + a1.y * b1.y;
    }
    if (q < nq) {
        const double2 a0 = row[q], b0 = xv[q];
        s0 += a0.x * b0.x + a0.y * b0.y;
    }
    const double s = wave_sum(s0 + s1);
    if (lane == 0) out[(long)blockIdx.y * so + i] = s;
}

// out[k] = sum_{i >= k} A[i][k] x[i] (A lower triangular, i.e. A^T x), HBM-read bound, in two launches:
// a workgroup owns 64 columns x GEMVT_ROWS rows (its 4 waves take rows v, v + 4, ...; each row read is one
// 512 B line; workgroups above the diagonal exit at once) and leaves a partial sum per column; the second
// kernel adds the row chunks in a fixed order (deterministic).  (One launch with a workgroup per 64 columns
// and all rows kept only 64 CUs busy: 75 us at N = 4096 against 22 us.)
constexpr int GEMVT_ROWS = 256;
// grid (Np/64, ceil(Np/GEMVT_ROWS), batch), 256 threads; part: [batch][chunks][Np]
__global__ void __launch_bounds__(256) gemv_lowerT_part_kernel(const double* __restrict__ A, const double* __restrict__ x,
                                                               double* __restrict__ part, int Np, long sA, long sx,
                                                               long sPart, const int* __restrict__ zmap = nullptr) {
    // a lane owns TWO adjacent columns (16-byte loads, 1 KB per wave and row); the strictly upper entries it may touch
    // inside the diagonal block are stored zeros.  grid (Np/128, chunks, batch).
    __shared__ double red[4][128];
    const int lane = threadIdx.x & 63, v = threadIdx.x >> 6, k0 = blockIdx.x * 128, k = k0 + 2 * lane;
    const int r0 = blockIdx.y * GEMVT_ROWS, r1 = min(Np, r0 + GEMVT_ROWS);
    const long bz = zmap ? zmap[blockIdx.z] : (int)blockIdx.z;
    double* __restrict__ po = part + bz * sPart + (long)blockIdx.y * Np;
    const bool live = k < Np;                           // (Np is a multiple of 64, not of 128)
    if (r1 <= k0) {                                     // entirely above the diagonal
        if (v == 0 && live) { po[k] = 0.0; po[k + 1] = 0.0; }
        return;
    }
    const double* __restrict__ Ab = A + bz * sA;
    const double* __restrict__ xv = x + bz * sx;
    double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0;
    int i = max(r0, k0) + v;
    const double2 zero2 = {0.0, 0.0};
    for (; i + 4 < r1; i += 8) {
        const double2 a = live ? *reinterpret_cast<const double2*>(Ab + (long)i * Np + k) : zero2;
        const double2 b = live ? *reinterpret_cast<const double2*>(Ab + (long)(i + 4) * Np + k) : zero2;
        const double xa = xv[i], xb = xv[i + 4];
        s0 += a.x * xa; s1 += a.y * xa;
        t0 += b.x * xb; t1 += b.y * xb;
    }
    if (i < r1) {
        const double2 a = live ? *reinterpret_cast<const double2*>(Ab + (long)i * Np + k) : zero2;
        s0 += a.x * xv[i]; s1 += a.y * xv[i];
    }
    red[v][2 * lane] = s0 + t0;
    red[v][2 * lane + 1] = s1 + t1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int c = threadIdx.x;
        if (k0 + c < Np) po[k0 + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}
// grid (Np/256, batch), 256 threads
__global__ void __launch_bounds__(256) gemv_lowerT_finish_kernel(const double* __restrict__ part, double* __restrict__ out,
                                                                 int Np, int chunks, long sPart, long so,
                                                                 const int* __restrict__ zmap = nullptr) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= Np) return;
    const long by = zmap ? zmap[blockIdx.y] : (int)blockIdx.y;
    const double* __restrict__ p = part + by * sPart + k;
    double t = 0.0;
    for (int c = k / GEMVT_ROWS; c < chunks; ++c) t += p[(long)c * Np];
    out[by * so + k] = t;
}

// out[b][row0 + r] = (yin ? yin[b][row0 + r] : 0) + sign * sum_{c < (tri ? r + 1 : ncols)} A[b][(row0 + r) ld + col0 + c] x[b][col0 + c]
// for r < nrows: the two products of a step of the blocked forward substitution w = L^-1 y from L and the inverses I_i of its
// diagonal blocks (fwd_subst below).  A wave per row, lanes along the row (16-byte loads), fixed reduction order.
// grid (nrows / 4, batch), 256 threads; row0, col0, ncols even (they are multiples of 64).
__global__ void __launch_bounds__(256) rowdot_kernel(const double* __restrict__ A, long ld, long sA, int row0, int nrows, int col0, int ncols,
                                                     int tri, const double* __restrict__ x, long sx, const double* __restrict__ yin,
                                                     long sy, double* __restrict__ out, long so, double sign) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const double2* __restrict__ row = reinterpret_cast<const double2*>(A + (long)blockIdx.y * sA + (long)(row0 + r) * ld + col0);
    const double2* __restrict__ xv = reinterpret_cast<const double2*>(x + (long)blockIdx.y * sx + col0);
    const int kend = tri ? r + 1 : ncols;                // (a lower-triangular operand stores exact zeros beyond its diagonal)
    const int nq = (kend + 1) >> 1;
    double s0 = 0.0, s1 = 0.0;
    for (int q = lane; q < nq; q += 64) {
        const double2 a = row[q], b = xv[q];
        s0 = fma(a.x, b.x, s0);
        s1 = fma(a.y, b.y, s1);
    }
    const double s = wave_sum(s0 + s1);
    if (lane == 0) out[(long)blockIdx.y * so + row0 + r] = (yin ? yin[(long)blockIdx.y * sy + row0 + r] : 0.0) + sign * s;
}

// a7 tail: nll[a] = 1/2 w^T w + sum_i log|L_ii| with w = L^-1 y (= 1/2 y^T alpha + 1/2 logdet K,
// optimize.py:352-355).  grid (batch), 256 threads, fixed-order reduction (deterministic).
__global__ void __launch_bounds__(256) nll_reduce_kernel(const double* __restrict__ L, const double* __restrict__ w,
                                                         double* __restrict__ nll, int N, int Np) {
    const int a = blockIdx.x, tid = threadIdx.x;
    __shared__ double red[4];
    const double* La = L + (long)a * Np * Np;
    const double* wa = w + (long)a * Np;
    double s = 0.0;
    for (int i = tid; i < N; i += 256) s += 0.5 * wa[i] * wa[i] + log(fabs(La[(long)i * Np + i]));
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) nll[a] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Analytic NLL gradient, Rasmussen & Williams eq. 5.9 (no reference function; optimize.py:371-375):
//   g_theta = 1/2 sum_ij W_ij dK_ij/dtheta,  W = K^-1 - alpha alpha^T,
//   dK/d ell_dd = Kse_ij (x_id - x_jd)^2 / ell_dd^3,  dK/d sf = 2 Kse / sf,  dK/d sn = 2 sn I.
// One pass over the lower triangle of K^-1 with Kse recomputed on the fly (HBM-read bound:
// 4 N^2 bytes); per-tile partial sums are written out and reduced in a fixed order.
// grid (Np/64, Np/64, batch), 256 threads; partial[b][tile][d+2]; batch element b: hyper row b, K^-1 and alpha number b
// (the lock-step restart search evaluates many hyper-parameter points of ONE data set at once).
__global__ void __launch_bounds__(256) nll_grad_kernel(const double* __restrict__ XT, const double* __restrict__ hyper,
                                                       const double* __restrict__ invK, const double* __restrict__ alpha,
                                                       double* __restrict__ partial, int N, int Np, int d,
                                                       const int* __restrict__ zmap = nullptr) {
    const int tn = blockIdx.x, tm = blockIdx.y, tid = threadIdx.x;
    const int tiles = Np / 64;
    const int bz = zmap ? zmap[blockIdx.z] : (int)blockIdx.z;   // (zmap: the matrices of a subset of a batch)
    hyper += (long)bz * (d + 2);
    invK += (long)bz * Np * Np;
    alpha += (long)bz * Np;
    double* out = partial + ((long)bz * tiles * tiles + (long)tm * tiles + tn) * (DMAX + 2);
    if (tn > tm) return;
    __shared__ double Xr[DMAX][64], Xc[DMAX][64], w[DMAX], ar[64], ac[64], red[4][DMAX + 2];
    const int m0 = tm * 64, n0 = tn * 64;
    for (int idx = tid; idx < 64 * d; idx += 256) {
        const int dd = idx >> 6, i = idx & 63;
        Xr[dd][i] = XT[(long)dd * Np + m0 + i];
        Xc[dd][i] = XT[(long)dd * Np + n0 + i];
    }
    if (tid < 64) { ar[tid] = alpha[m0 + tid]; ac[tid] = alpha[n0 + tid]; }
    if (tid < DMAX) w[tid] = (tid < d) ? 1.0 / (hyper[tid] * hyper[tid]) : 0.0;
    const double sf2 = hyper[d] * hyper[d];
    __syncthreads();
    double g[DMAX], gsf = 0.0, gtr = 0.0;
#pragma unroll
    for (int dd = 0; dd < DMAX; ++dd) g[dd] = 0.0;
    for (int s = 0; s < 16; ++s) {
        const int idx = tid + 256 * s, r = idx >> 6, c = idx & 63;
        const int i = m0 + r, j = n0 + c;
        if (i < N && j <= i) {
            const double Wij = invK[(long)i * Np + j] - ar[r] * ac[c];
            const double mult = (j < i) ? 2.0 : 1.0;  // symmetric counterpart
            double dist = 0.0, df2[DMAX];
#pragma unroll
            for (int dd = 0; dd < DMAX; ++dd) {
                const double df = (dd < d) ? Xr[dd][r] - Xc[dd][c] : 0.0;
                df2[dd] = df * df;
                dist += df2[dd] * w[dd];
            }
            const double wk = mult * Wij * sf2 * exp(-0.5 * dist);
#pragma unroll
            for (int dd = 0; dd < DMAX; ++dd) g[dd] += wk * df2[dd];
            gsf += wk;
            if (i == j) gtr += Wij;
        }
    }
#pragma unroll
    for (int dd = 0; dd < DMAX; ++dd) {
        const double s = wave_sum(g[dd]);
        if ((tid & 63) == 0) red[tid >> 6][dd] = s;
    }
    gsf = wave_sum(gsf);
    gtr = wave_sum(gtr);
    if ((tid & 63) == 0) { red[tid >> 6][DMAX] = gsf; red[tid >> 6][DMAX + 1] = gtr; }
    __syncthreads();
    if (tid < DMAX + 2) out[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// final reduction of the gradient partials -> grad[d+2]: one wave per parameter in turn, lanes stride the lower tiles,
// butterfly sum (a fixed order: deterministic).  (The first version let d + 2 threads walk all (Np/64)^2 / 2 partials one
// after the other: 0.47 ms of dependent loads at N = 4096, a fifth of an NLL + gradient evaluation.)
// grid (batch): element b reads partial[b], hyper row b and writes grad + b * gstride
__global__ void __launch_bounds__(256) nll_grad_finish_kernel(const double* __restrict__ partial,
                                                              const double* __restrict__ hyper,
                                                              double* __restrict__ grad, int Np, int d, int gstride = 0,
                                                              const int* __restrict__ zmap = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tiles = Np / 64;
    const int bx = zmap ? zmap[blockIdx.x] : (int)blockIdx.x;
    const long nt = (long)tiles * tiles;
    partial += (long)bx * nt * (DMAX + 2);
    hyper += (long)bx * (d + 2);
    grad += (long)bx * gstride;
    for (int e = wave; e < d + 2; e += 4) {
        const int col = e < d ? e : (e == d ? DMAX : DMAX + 1);
        double s = 0.0;
        for (long t = lane; t < nt; t += 64) {
            const int tm = (int)(t / tiles), tn = (int)(t % tiles);
            if (tn <= tm) s += partial[t * (DMAX + 2) + col];
        }
        s = wave_sum(s);
        if (lane == 0) {
            double g;
            if (e < d) g = 0.5 * s / (hyper[e] * hyper[e] * hyper[e]);
            else if (e == d) g = 0.5 * s * 2.0 / hyper[d];
            else g = 0.5 * s * 2.0 * hyper[d + 1];
            grad[e] = g;
        }
    }
}

// mirror the strictly-lower triangle into the upper one (K^-1 is exported as a full symmetric
// matrix, gp_class.py:698).  grid (Np/64, Np/64, batch); 64 x 64 tiles through LDS.
__global__ void __launch_bounds__(256) symmetrize_kernel(double* __restrict__ A, int Np) {
    const int tn = blockIdx.x, tm = blockIdx.y;
    if (tn > tm) return;
    __shared__ double tile[64][65];
    double* Aa = A + (long)blockIdx.z * Np * Np;
    const int tid = threadIdx.x, m0 = tm * 64, n0 = tn * 64;
    for (int idx = tid; idx < 4096; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        tile[r][c] = Aa[(long)(m0 + r) * Np + n0 + c];
    }
    __syncthreads();
    for (int idx = tid; idx < 4096; idx += 256) {
        const int r = idx >> 6, c = idx & 63;  // writes element (n0 + r, m0 + c) = tile[c][r]
        if (tm != tn || c > r) Aa[(long)(n0 + r) * Np + m0 + c] = tile[c][r];
    }
}

// XT = X^T for a lower-triangular X (= L^-1): the 64 x 64 blocks on and below the diagonal of X become the blocks on and
// above the diagonal of XT (diagonal blocks whole, zeros included); nothing else of XT is written or ever read.  Makes the
// operands of K^-1 = X^T X contiguous along the contraction index (vargemm_persist.hpp, PG_XTX).  grid (Np/64, Np/64, batch).
__global__ void __launch_bounds__(256) transpose_lower_kernel(const double* __restrict__ X, double* __restrict__ XT, int Np,
                                                              const int* __restrict__ zmap = nullptr) {
    const int bj = blockIdx.x, bi = blockIdx.y;
    if (bj > bi) return;
    __shared__ double tile[64][65];
    const long bz = zmap ? zmap[blockIdx.z] : (int)blockIdx.z;
    const double* Xa = X + bz * Np * Np;
    double* Ta = XT + bz * Np * Np;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 4096; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        tile[r][c] = Xa[(long)(bi * 64 + r) * Np + bj * 64 + c];
    }
    __syncthreads();
    for (int idx = tid; idx < 4096; idx += 256) {
        const int r = idx >> 6, c = idx & 63;  // XT element (bj*64 + r, bi*64 + c) = X(bi*64 + c, bj*64 + r)
        Ta[(long)(bj * 64 + r) * Np + bi * 64 + c] = tile[c][r];
    }
}

// f64 MFMA fragment-layout probe + rate micro-benchmark (gpmpc_mfma_selftest).
__global__ void __launch_bounds__(64) mfma_probe_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                        double* __restrict__ D) {
    const int l = threadIdx.x;
    d4 c = d4{0.0, 0.0, 0.0, 0.0};
    c = mfma16(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c);
#pragma unroll
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

// (min 4 waves per SIMD in the launch bounds = at most 128 registers = accumulators stay in arch VGPRs: with
// the default bounds the compiler keeps them in AGPRs and copies 64 registers in and out of the loop body on
// every iteration, which once made this benchmark report 45 instead of 77 TFLOP/s)
__global__ void __launch_bounds__(256, 4) mfma_rate_kernel(double* __restrict__ out, int iters) {
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    d4 c0 = d4{0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = mfma16(a, b, c0);
        c1 = mfma16(a, b, c1);
        c2 = mfma16(a, b, c2);
        c3 = mfma16(a, b, c3);
    }
    out[(long)blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

}  // namespace gpmpc
