// Exact moment matching (Deisenroth) -- a11, gp_exact_moment gp_functions.py:344-418 + maha :421-430,
// zero prior mean, restated for the device without ever materialising the N x N matrix Q:
//
//   per output a:   beta_a = K_a^-1 y_a (:383),  iR = (Sigma + Lambda_a)^-1 (:385-387),
//                   q_i = sf_a^2 prod(ell_a) det(Sigma + Lambda_a)^-1/2 exp(-1/2 v_i iR v_i^T) (:388-391),
//                   mean_a = sum_i q_i beta_ai (:392-393),  log_k[i,a] = log sf_a^2 - 1/2 sum_d (v_id/ell_ad)^2 (:394-396)
//   per pair b<=a:  R = Sigma diag(1/ell_a^2 + 1/ell_b^2) + I,  t = det(R)^-1/2,  S = R^-1 Sigma / 2 (:402-407),
//                   Q_ij = exp(log_k[i,a] + log_k[j,b] + maha(ii, -ij, S)_ij),  ii = v/ell_a^2, ij = v/ell_b^2,
//                   cov_ab = t sum_ij (beta_ai beta_bj - [a==b] K_a^-1_ij) Q_ij (:408-414)
//   cov_aa += sf_a^2 (:415),  cov -= mean mean^T (:416).
//
// maha(ii, -ij, S)_ij = ii_i S ii_i^T + ij_j S ij_j^T + 2 (ii_i S) . ij_j, so a tile of Q needs one
// d-vector dot product and one exp per entry.  The dot products of a 16 x 16 tile are a depth-8 product
// on the matrix pipe (two v_mfma_f64_16x16x4_f64); the VALU is left with one lean exp and the weighted
// accumulation per entry: exp/VALU bound for a != b, additionally one read of the lower triangle of
// K_a^-1 for a == b (symmetry halves those pairs).
#pragma once
#include "gp_kernels.hpp"

namespace gpmpc {

// ---- tiny dense helpers on d x d matrices (d <= DMAX), one thread each ----------------------------
// Gaussian elimination with partial pivoting: solves A X = Bm (nrhs columns) in place, returns |det A|
// (the reference's `determinant` is exp(trace(log(R_qr))) = |det|, gp_functions.py:378-380).
__device__ inline double small_solve(double* A, double* Bm, int d, int nrhs) {
    double det = 1.0;
    for (int k = 0; k < d; ++k) {
        int piv = k;
        double best = fabs(A[k * d + k]);
        for (int r = k + 1; r < d; ++r)
            if (fabs(A[r * d + k]) > best) { best = fabs(A[r * d + k]); piv = r; }
        if (piv != k) {
            for (int c = 0; c < d; ++c) { const double t = A[k * d + c]; A[k * d + c] = A[piv * d + c]; A[piv * d + c] = t; }
            for (int c = 0; c < nrhs; ++c) { const double t = Bm[k * nrhs + c]; Bm[k * nrhs + c] = Bm[piv * nrhs + c]; Bm[piv * nrhs + c] = t; }
        }
        const double pv = A[k * d + k];
        det *= fabs(pv);
        for (int r = k + 1; r < d; ++r) {
            const double f = A[r * d + k] / pv;
            for (int c = k; c < d; ++c) A[r * d + c] -= f * A[k * d + c];
            for (int c = 0; c < nrhs; ++c) Bm[r * nrhs + c] -= f * Bm[k * nrhs + c];
        }
    }
    for (int c = 0; c < nrhs; ++c)
        for (int r = d - 1; r >= 0; --r) {
            double s = Bm[r * nrhs + c];
            for (int k = r + 1; k < d; ++k) s -= A[r * d + k] * Bm[k * nrhs + c];
            Bm[r * nrhs + c] = s / A[r * d + r];
        }
    return det;
}

// Gauss-Jordan elimination with partial pivoting on the augmented matrix M = [A | B] (d rows, nc = d + nrhs columns,
// row stride GJ_LD) held in LDS, executed by the whole workgroup (one thread per entry): on return the B part holds
// A^-1 B; returns |det A| (same value in every thread).  `piv` is a 2-int LDS slot.
constexpr int GJ_LD = 2 * DMAX;
__device__ inline double gauss_jordan_lds(double* M, int d, int nc, int* piv) {
    const int tid = threadIdx.x, r = tid / GJ_LD, c = tid % GJ_LD;
    double det = 1.0;
    for (int k = 0; k < d; ++k) {
        if (tid == 0) {
            int best = k;
            double bv = fabs(M[k * GJ_LD + k]);
            for (int q = k + 1; q < d; ++q)
                if (fabs(M[q * GJ_LD + k]) > bv) { bv = fabs(M[q * GJ_LD + k]); best = q; }
            piv[0] = best;
        }
        __syncthreads();
        const int p = piv[0];
        if (p != k && r == 0 && c < nc) {                    // swap rows k and p (one thread per column)
            const double t = M[k * GJ_LD + c];
            M[k * GJ_LD + c] = M[p * GJ_LD + c];
            M[p * GJ_LD + c] = t;
        }
        __syncthreads();
        const double pv = M[k * GJ_LD + k];
        det *= fabs(pv);
        const double f = (r < d && r != k) ? M[r * GJ_LD + k] / pv : 0.0;
        const double rowk = (c < nc) ? M[k * GJ_LD + c] : 0.0;
        __syncthreads();
        if (r < d && c < nc) {
            if (r == k) M[r * GJ_LD + c] = rowk / pv;
            else M[r * GJ_LD + c] -= f * rowk;
        }
        __syncthreads();
    }
    return det;
}

// per-input small algebra.  Layout of `prep` per input b (doubles):
//   [a < Ny]       iR_a[d*d], c_a                              -> Ny * (d*d + 1)
//   [pair p]       S_p[d*d], t_p                               -> P  * (d*d + 1),  p = a(a+1)/2 + b
// grid (B * (Ny + P)), DMAX * GJ_LD threads: one workgroup per (input, item), its d x d systems solved in LDS.
// (A first version gave each item ONE thread with its matrices in scratch memory: 0.28 ms per input at C3, an
//  eighth of the whole exact-moment step, all of it latency.)
__global__ void __launch_bounds__(DMAX * GJ_LD) em_prep_kernel(const double* __restrict__ hyper, const double* __restrict__ Sigma,
                                                               double* __restrict__ prep, int B, int Ny, int d,
                                                               unsigned long long* __restrict__ bnd = nullptr) {
    const int P = Ny * (Ny + 1) / 2, items = Ny + P;
    const int b = blockIdx.x / items, it = blockIdx.x % items;
    const double* Sg = Sigma + (long)b * d * d;
    const int stride = d * d + 1;
    double* out = prep + ((long)b * items + it) * stride;
    __shared__ double M[DMAX * GJ_LD];
    __shared__ int piv[2];
    const int tid = threadIdx.x, r = tid / GJ_LD, c = tid % GJ_LD;
    if (it < Ny) {
        const double* hy = hyper + (long)it * (d + 2);
        // iR = iLambda (I - (I + Sigma iLambda)^-1 (Sigma iLambda)),  R = Sigma + Lambda
        if (r < d && c < 2 * d) {
            const int cc = c < d ? c : c - d;
            const double sil = Sg[r * d + cc] / (hy[cc] * hy[cc]);
            M[r * GJ_LD + c] = c < d ? sil + (r == cc ? 1.0 : 0.0) : sil;      // [I + Sigma iLambda | Sigma iLambda]
        }
        __syncthreads();
        gauss_jordan_lds(M, d, 2 * d, piv);
        if (r < d && c < d) out[r * d + c] = ((r == c ? 1.0 : 0.0) - M[r * GJ_LD + d + c]) / (hy[r] * hy[r]);
        __syncthreads();
        if (r < d && c < d) M[r * GJ_LD + c] = Sg[r * d + c] + (r == c ? hy[r] * hy[r] : 0.0);
        __syncthreads();
        const double det = gauss_jordan_lds(M, d, d, piv);
        if (tid == 0) {
            double prod = 1.0;
            for (int q = 0; q < d; ++q) prod *= hy[q];
            out[d * d] = hy[d] * hy[d] / sqrt(det) * prod;
        }
    } else {
        int p = it - Ny, a = 0;
        while ((a + 1) * (a + 2) / 2 <= p) ++a;
        const int bb = p - a * (a + 1) / 2;
        const double* ha = hyper + (long)a * (d + 2);
        const double* hb = hyper + (long)bb * (d + 2);
        if (bnd && tid < 4) bnd[((long)b * P + p) * 4 + tid] = 0ull;       // (operand magnitudes of this pair: em_operands_kernel)
        if (r < d && c < 2 * d) {
            const int cc = c < d ? c : c - d;
            M[r * GJ_LD + c] = c < d ? Sg[r * d + cc] * (1.0 / (ha[cc] * ha[cc]) + 1.0 / (hb[cc] * hb[cc])) + (r == cc ? 1.0 : 0.0)
                                     : Sg[r * d + cc] * 0.5;                     // [R2 | Sigma / 2]
        }
        __syncthreads();
        const double det = gauss_jordan_lds(M, d, 2 * d, piv);
        if (r < d && c < d) out[r * d + c] = M[r * GJ_LD + d + c];
        if (tid == 0) out[d * d] = 1.0 / sqrt(det);
    }
}

// mean_a = sum_i q_i beta_ai.  grid (Ny, B, EM_MEAN_CHUNKS), 256 threads: partial sums per chunk of the training
// points (six workgroups for the whole sum took 0.2 ms at C3), added in a fixed order by em_mean_finish_kernel.
constexpr int EM_MEAN_CHUNKS = 16;
__global__ void __launch_bounds__(256) em_mean_kernel(const double* __restrict__ XT, const double* __restrict__ Z,
                                                      const double* __restrict__ beta, const double* __restrict__ prep,
                                                      double* __restrict__ mpart, int N, int Np, int d, int Ny) {
    const int a = blockIdx.x, b = blockIdx.y, ch = blockIdx.z, tid = threadIdx.x;
    const int P = Ny * (Ny + 1) / 2, stride = d * d + 1;
    const double* pr = prep + ((long)b * (Ny + P) + a) * stride;
    __shared__ double iR[DMAX * DMAX], mu[DMAX], red[4];
    for (int e = tid; e < d * d; e += 256) iR[e] = pr[e];
    if (tid < d) mu[tid] = Z[(long)b * d + tid];
    __syncthreads();
    const double c = pr[d * d];
    const int clen = (N + EM_MEAN_CHUNKS - 1) / EM_MEAN_CHUNKS, ibeg = ch * clen, iend = min(N, ibeg + clen);
    double s = 0.0;
    for (int i = ibeg + tid; i < iend; i += 256) {
        double v[DMAX];
        for (int k = 0; k < d; ++k) v[k] = XT[(long)k * Np + i] - mu[k];
        double qf = 0.0;
        for (int r = 0; r < d; ++r) {
            double t = 0.0;
            for (int k = 0; k < d; ++k) t += v[k] * iR[k * d + r];     // T = v iR
            qf += t * v[r];
        }
        s += c * exp(-0.5 * qf) * beta[(long)a * Np + i];
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) mpart[((long)b * Ny + a) * EM_MEAN_CHUNKS + ch] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(256) em_mean_finish_kernel(const double* __restrict__ mpart, double* __restrict__ mean,
                                                             int count) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    double s = 0.0;
    for (int ch = 0; ch < EM_MEAN_CHUNKS; ++ch) s += mpart[(long)e * EM_MEAN_CHUNKS + ch];
    mean[e] = s;
}

// Cross-term depth of the matrix-pipe path: the value kernels below exist for KD = 8 (d <= 8: two 16x16x4 steps per
// tile) and KD = 16 (d <= 16 = DMAX: four); the derivative kernels further down for depth EMK = 8 only.
constexpr int em_depth(int d) { return d <= 8 ? 8 : 16; }
constexpr double EM_PAD_LOG = -1.0e5;    // row log-weight of padded points (em_operands_kernel)

// Magnitudes of a pair's operands, four words per (input, pair): bit patterns of max |La|, max |Lb|, max |U|, max |Wt| over the
// points (em_operands_kernel: atomicMax on the patterns, which order like the non-negative doubles they are; NaN sorts above
// inf).  An exponent of the pair sums is La_i + Lb_j + sum_k U_ik Wt_jk, so |c| <= max|La| + max|Lb| + KD max|U| max|Wt|: when
// that stays below 1e9 the table exp needs no clamp (exp_tab, gp_kernels.hpp).  em_prep_kernel zeroes the words.  The pair-sum
// kernels exist with and without the clamp (+1.8 % on the C3 step when always on); both are launched and the workgroups of the
// one that is not needed leave at once -- a test point 1e5 length scales away, or sf = 0, gives exact zeros instead of NaN.
__device__ __forceinline__ bool em_needs_clamp(const unsigned long long* __restrict__ w, int KD) {
    const double mLa = __longlong_as_double((long long)w[0]), mLb = __longlong_as_double((long long)w[1]);
    const double mU = __longlong_as_double((long long)w[2]), mW = __longlong_as_double((long long)w[3]);
    const double bound = mLa + mLb + (double)KD * mU * mW;
    return !(bound < 1.0e9);                                     // (NaN or inf anywhere: clamp)
}
__device__ __forceinline__ double em_wave_max_pattern(double v) {   // v >= 0 or NaN; the lane with the largest bit pattern wins
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double o = __shfl_xor(v, m);
        if ((unsigned long long)__double_as_longlong(o) > (unsigned long long)__double_as_longlong(v)) v = o;
    }
    return v;
}

// Per-(input, pair, point) operands of the pair kernel.  One thread per (b, p, i); arrays are
// [b][p][...][Np] so that the pair kernel's loads are contiguous in the point index:
//   U[k][i]  = (ii_i S)_k * 2        (A operand of the cross-term product, k < EMK, zero padded)
//   Wt[k][j] = ij_j,k                 (B operand)
//   La[i] = log_k[i,a] + ii_i S ii_i^T,   Lb[j] = log_k[j,b] + ij_j S ij_j^T
// (gp_functions.py:394-396, :400-408 with maha expanded as in the header comment).
template <int KD>
__global__ void __launch_bounds__(256) em_operands_kernel(const double* __restrict__ XT, const double* __restrict__ Z,
                                                          const double* __restrict__ hyper, const double* __restrict__ prep,
                                                          double* __restrict__ ops, int N, int Np, int d, int Ny,
                                                          unsigned long long* __restrict__ bnd) {
    constexpr int EMK = KD;
    const int P = Ny * (Ny + 1) / 2, stride = d * d + 1;
    const int i = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y, b = blockIdx.z;
    if (i >= Np) return;                                         // (whole waves: Np is a multiple of 64)
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= p) ++a;
    const int bb = p - a * (a + 1) / 2;
    const double* S = prep + ((long)b * (Ny + P) + Ny + p) * stride;
    const double* ha = hyper + (long)a * (d + 2);
    const double* hb = hyper + (long)bb * (d + 2);
    double* o = ops + ((long)b * P + p) * (2 * EMK + 2) * Np;
    double v[DMAX], ii[DMAX], ij[DMAX];
    double lka = 0.0, lkb = 0.0;
    for (int k = 0; k < d; ++k) {
        v[k] = (i < N) ? XT[(long)k * Np + i] - Z[(long)b * d + k] : 0.0;
        ii[k] = v[k] / (ha[k] * ha[k]);
        ij[k] = v[k] / (hb[k] * hb[k]);
        lka += v[k] * v[k] / (ha[k] * ha[k]);
        lkb += v[k] * v[k] / (hb[k] * hb[k]);
    }
    double qa = 0.0, qb = 0.0, mU = 0.0, mW = 0.0;
    for (int c = 0; c < EMK; ++c) {
        double ua = 0.0, ub = 0.0;
        if (c < d)
            for (int k = 0; k < d; ++k) { ua += ii[k] * S[k * d + c]; ub += ij[k] * S[k * d + c]; }
        o[(long)c * Np + i] = 2.0 * ua;
        o[(long)(EMK + c) * Np + i] = (c < d) ? ij[c] : 0.0;
        if (c < d) {
            qa += ua * ii[c]; qb += ub * ij[c];
            const double au = fabs(2.0 * ua), aw = fabs(ij[c]);
            if (!(au <= mU)) mU = au;                            // (keeps a NaN)
            if (!(aw <= mW)) mW = aw;
        }
    }
    // (padded points: a log-weight that makes every Q of their ROW an exact zero through any of the exps -- ldexp underflows,
    //  the table exps' integer part stays inside 32 bits down to -7e5 --, so that the a == b sums need no row mask on K^-1,
    //  whose padded rows are identity rows)
    const double lai = i < N ? (2.0 * log(ha[d]) - 0.5 * lka) + qa : EM_PAD_LOG, lbi = (2.0 * log(hb[d]) - 0.5 * lkb) + qb;
    o[(long)(2 * EMK) * Np + i] = lai;
    o[(long)(2 * EMK + 1) * Np + i] = lbi;
    // the pair's operand magnitudes (em_needs_clamp): one atomic per wave and word
    const double wLa = em_wave_max_pattern(fabs(lai)), wLb = em_wave_max_pattern(fabs(lbi));
    const double wU = em_wave_max_pattern(mU), wW = em_wave_max_pattern(mW);
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* w = bnd + ((long)b * P + p) * 4;
        atomicMax(w + 0, (unsigned long long)__double_as_longlong(wLa));
        atomicMax(w + 1, (unsigned long long)__double_as_longlong(wLb));
        atomicMax(w + 2, (unsigned long long)__double_as_longlong(wU));
        atomicMax(w + 3, (unsigned long long)__double_as_longlong(wW));
    }
}

// Pair sums on a 64-row strip.  256 threads = 4 waves; wave w owns rows 16w..16w+15 of the strip and sweeps column tiles
// (64 wide) without any barrier inside a tile: the cross terms 2 (ii_i S) . ij_j of a 16 x 16 tile are two
// v_mfma_f64_16x16x4_f64 (depth EMK = 8, zero padded; four at EMK = 16), the VALU takes the exp and accumulates
// (beta_ai beta_bj - [a==b] K^-1_ij) Q_ij.  For a == b the summand is symmetric in (i, j): only column tiles up to the diagonal
// are visited, off-diagonal tiles counted twice.
// grid (tiles * nch, P, B): workgroup (strip ti, chunk c) sweeps the column tiles [c chunk, (c+1) chunk) of its strip
// (r05: one workgroup per strip left the a == b launch at the mercy of its longest strip -- 128 tiles against a mean of 64.5 --
// and the a != b launch with 2.5 rounds of 128-tile workgroups on the chip's 768 slots, i.e. a half-empty last round;
// profiles/r05_kernel_trace_bench_c3.txt: 482 + 766 us per input at C3); partial[(b*P + p) * tiles * nch + blockIdx.x],
// chunks beyond a strip's range write 0.
// r05: the VALU work per entry cut from 20.8 (r04's em_pair_kernel) to 14.8 instructions (the kernel is bound by VALU issue,
// profiles/r04_pmc_em_tab*, r05_pmc_em_*):
//  * the matrix instruction's accumulator is INITIALISED with La_i + Lb_j, so the exp argument comes straight out of the
//    matrix pipe (one add per entry instead of two), and the kernel is held to <= 168 registers (three waves per SIMD at
//    least), which makes the compiler keep the accumulators in VGPRs: the AGPR form cost two v_accvgpr_read per entry;
//  * sum_ij beta_i beta_j Q_ij = sum_i beta_i (sum_j beta_j Q_ij): one fma per entry into a per-row accumulator, beta_i at the
//    end; for a == b a second accumulator takes K^-1_ij Q_ij, and the factor 2 of the off-diagonal tiles is applied once
//    (everything accumulated before the diagonal tile is doubled) instead of per entry;
//  * TAB = 2: exp through the 32-entry table that meets every LDS bank once (exp_tab32, gp_kernels.hpp), TAB = 1: the
//    2048-entry table of r04, TAB = 0: the polynomial exp_lean.
template <bool DIAG, int KD, int TAB, bool CLAMP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
em_pair2_kernel(const double* __restrict__ ops, const double* __restrict__ beta, const double* __restrict__ invK,
                double* __restrict__ partial, int N, int Np, int Ny, int crow_mode, const double* __restrict__ etab, int chunk,
                int slot_stride, const unsigned long long* __restrict__ bnd) {
    constexpr int EMK = KD;
    constexpr int NQ = ((KD + 2) * 64 + 255) / 256;
    const int P = Ny * (Ny + 1) / 2, tiles = Np / 64, nch = (tiles + chunk - 1) / chunk;
    const int ti = blockIdx.x / nch, p = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= p) ++a;
    const int bb = p - a * (a + 1) / 2;
    if ((a == bb) != DIAG) return;
    if (em_needs_clamp(bnd + ((long)b * P + p) * 4, KD) != CLAMP) return;      // (the other instantiation serves this pair)
    // (slot_stride >= tiles * nch partial sums per pair: the a == b launch may use more of them, em_diag_kernel)
    const int jt_beg = (blockIdx.x % nch) * chunk, jt_end = ti < tiles ? min(DIAG ? ti + 1 : tiles, jt_beg + chunk) : 0;
    double* __restrict__ pout = partial + ((long)b * P + p) * slot_stride + blockIdx.x;
    if (jt_beg >= jt_end) {                      // (a chunk beyond the diagonal of an a == b strip, or an unused slot)
        if (tid == 0) *pout = 0.0;
        return;
    }
    const double* __restrict__ o = ops + ((long)b * P + p) * (2 * EMK + 2) * Np;
    const double* __restrict__ Wt = o + (long)EMK * Np;
    const double* __restrict__ La = o + (long)(2 * EMK) * Np;
    const double* __restrict__ Lb = o + (long)(2 * EMK + 1) * Np;
    const double* __restrict__ ba = beta + (long)a * Np;
    const double* __restrict__ bbv = beta + (long)bb * Np;
    const double* __restrict__ iK = invK + (long)a * Np * Np;
    __shared__ double red[4];
    __shared__ double Cs[2][EMK + 2][64];
    __shared__ double Et[TAB == 1 ? EXPT_N : (TAB == 2 ? EXPT32_N : 1)];
    if (TAB == 1) exp_tab_fill(Et, etab, tid, 256);              // (visible behind the barrier that follows the first stage())
    if (TAB == 2) exp_tab32_fill(Et, tid);
    const int fr = lane & 15, fk = lane >> 4, i0 = ti * 64 + 16 * wave;
    double af[KD / 4];
#pragma unroll
    for (int s4 = 0; s4 < KD / 4; ++s4) af[s4] = o[(long)(4 * s4 + fk) * Np + i0 + fr];
    double la[4], bai[4], racc[4];
    int irow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        irow[r] = i0 + crow(lane, r, crow_mode);
        la[r] = La[irow[r]];
        bai[r] = (irow[r] < N) ? ba[irow[r]] : 0.0;
        racc[r] = 0.0;
    }
    double kacc = 0.0;
    double st[NQ];
    auto fetch = [&](int jt) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid + 256 * q, rw = e >> 6, cl = e & 63, j = jt * 64 + cl;
            double v = 0.0;
            if (rw < EMK) v = Wt[(long)rw * Np + j];
            else if (rw == EMK) v = Lb[j];
            else if (rw == EMK + 1) v = (j < N) ? bbv[j] : 0.0;
            st[q] = v;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid + 256 * q, rw = e >> 6, cl = e & 63;
            if (rw < EMK + 2) Cs[buf][rw][cl] = st[q];
        }
    };
    fetch(jt_beg);
    stage(0);
    __syncthreads();
    int cur = 0;
    // a == b: K^-1 of a 16 x 16 sub-tile travels ONE SUB-TILE AHEAD of its use (r06): `ikn` is loaded while the previous
    // sub-tile is computed, `ikc` is what the current one multiplies.  (r01-r05 loaded the tile's sixteen values at the top of
    // the tile and the first sub-tile waited for all of them -- and, through the same counter, for the next tile's operands:
    // one exposed HBM latency per tile and wave, 461 us per input at C3 against 134 us of VALU work and 290 us of K^-1 at
    // 5.5 TB/s.)  The row mask is applied when the values change registers; padded rows are inside the allocation.
    double ikc[4], ikn[4];
    long kofs[4];
    bool rok[4];
    if (DIAG) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rok[r] = irow[r] < N;
            kofs[r] = (long)irow[r] * Np + jt_beg * 64 + fr;
            ikn[r] = iK[kofs[r]];
            kofs[r] += 16;
        }
    }
    for (int jt = jt_beg; jt < jt_end; ++jt) {
        if (!DIAG && jt + 1 < jt_end) fetch(jt + 1);
        if (DIAG) {
            if (jt == ti) {                     // everything so far came from tiles below the diagonal: counted twice
#pragma unroll
                for (int r = 0; r < 4; ++r) racc[r] *= 2.0;
                kacc *= 2.0;
            }
        }
        // one 16 x 16 tile at a time (`unroll 1`, r05): four exps in flight per lane instead of sixteen -- 80 / 126 registers instead
        // of 156 / 190, six / four waves per SIMD instead of three / two, and the table look-ups' and the matrix results'
        // latencies hide behind other waves (C3, same box: EM phase 38.3 -> 36.7 ms; unroll 2: 37.0; profiles/r05_em_occupancy_ab.txt)
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            if (DIAG) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ikc[r] = rok[r] ? ikn[r] : 0.0;
                // (the next tile's operands are requested here, behind the wait the line above implies, not in front of it)
                if (t == 0 && jt + 1 < jt_end) fetch(jt + 1);
                if (t < 3 || jt + 1 < jt_end) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ikn[r] = iK[kofs[r]];
                        kofs[r] += 16;
                    }
                }
            }
            const int cl = 16 * t + fr;
            const double lbj = Cs[cur][EMK][cl];
            const double bj = Cs[cur][EMK + 1][cl];
            d4 c;
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = la[r] + lbj;
#pragma unroll
            for (int s4 = 0; s4 < KD / 4; ++s4) c = mfma16(af[s4], Cs[cur][4 * s4 + fk][cl], c);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // no per-entry masks: beta is zero in padded rows / columns, K^-1's padded rows are masked at the load
                // above and its padded columns are exact zeros in live rows (identity padding), Q is finite everywhere
                const double q = TAB == 2 ? exp_tab32(c[r], Et) : TAB == 1 ? exp_tab<CLAMP>(c[r], Et) : exp_lean(c[r]);
                racc[r] = fma(bj, q, racc[r]);
                if (DIAG) kacc = fma(ikc[r], q, kacc);
            }
        }
        if (jt + 1 < jt_end) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    double acc = (bai[0] * racc[0] + bai[1] * racc[1]) + (bai[2] * racc[2] + bai[3] * racc[3]);
    if (DIAG) {
        acc -= kacc;
        if (jt_end <= ti) acc *= 2.0;           // a chunk entirely below the diagonal: every tile of it counts twice
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) *pout = (red[0] + red[1]) + (red[2] + red[3]);
}

// The a == b pair sums on a BALANCED schedule (r06).  em_pair2_kernel<true> cuts the lower triangle of tiles by strips and
// chunks: workgroups of 1 .. 64 tiles, 1 152 of them at C3 for the chip's 1 024 places -- list scheduling of those lengths
// ends 1.56 x later than the balanced time, and the launch (461 us per input, 1.6 GB of K^-1 = 3.5 TB/s on average) spends its
// tail on a few long strips.  Here the T = tiles (tiles + 1) / 2 tiles of a pair are numbered row by row and cut in `segs`
// equal ranges (host: em_diag_default_segs, 256 per pair at C3); a workgroup walks its range, and when the range crosses a
// strip's diagonal tile it closes that strip (weights: tiles left of the diagonal count twice) and loads the next strip's row
// operands.  Tile body: as em_pair2_kernel, with K^-1 of a 16 x 16 sub-tile requested at the sub-tile's top and consumed by its
// last four instructions, unmasked (padded rows have Q = 0: EM_PAD_LOG).
// grid (slot_stride, Ny, B): partial[(b*P + p(a,a)) * slot_stride + blockIdx.x], workgroups >= segs write 0.
// (register budget: at depth 8 the kernel is held to 80 registers = six waves per SIMD, the a != b launch's figure, at the price
//  of 22 spilled registers outside the tile loop -- next to that launch what limits the pair is wave slots: C3 EM roll-out
//  108 registers / four waves 29.5 ms, 96 / five 29.4, 80 / six 29.1; profiles/r06_em_diag_maskless_ab.txt)
template <int KD, int TAB, bool CLAMP>
__global__ void __launch_bounds__(256, KD == 8 ? 6 : 3)
em_diag_kernel(const double* __restrict__ ops, const double* __restrict__ beta, const double* __restrict__ invK,
               double* __restrict__ partial, int N, int Np, int Ny, int crow_mode, const double* __restrict__ etab, int segs,
               int slot_stride, const unsigned long long* __restrict__ bnd) {
    constexpr int EMK = KD;
    constexpr int NQ = ((KD + 2) * 64 + 255) / 256;
    const int P = Ny * (Ny + 1) / 2, tiles = Np / 64;
    const int a = blockIdx.y, p = a * (a + 1) / 2 + a, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* __restrict__ pout = partial + ((long)b * P + p) * slot_stride + blockIdx.x;
    if (em_needs_clamp(bnd + ((long)b * P + p) * 4, KD) != CLAMP) return;      // (the other instantiation serves this pair)
    const long T = (long)tiles * (tiles + 1) / 2;
    const long lo = (int)blockIdx.x < segs ? T * blockIdx.x / segs : 0, hi = (int)blockIdx.x < segs ? T * (blockIdx.x + 1) / segs : 0;
    if (lo >= hi) {
        if (tid == 0) *pout = 0.0;
        return;
    }
    int ti = (int)((sqrt(8.0 * (double)lo + 1.0) - 1.0) * 0.5);     // strip and column tile of tile number lo
    while ((long)(ti + 1) * (ti + 2) / 2 <= lo) ++ti;
    while ((long)ti * (ti + 1) / 2 > lo) --ti;
    int jt = (int)(lo - (long)ti * (ti + 1) / 2);
    const double* __restrict__ o = ops + ((long)b * P + p) * (2 * EMK + 2) * Np;
    const double* __restrict__ Wt = o + (long)EMK * Np;
    const double* __restrict__ La = o + (long)(2 * EMK) * Np;
    const double* __restrict__ Lb = o + (long)(2 * EMK + 1) * Np;
    const double* __restrict__ ba = beta + (long)a * Np;
    // K^-1: accumulator register r of a lane sits crow-step rows below register 0 -- four wave-uniform row bases and ONE
    // 32-bit element offset per lane (host: Np^2 < 2^29), which the compiler turns into scalar-base loads
    const long cstep = (long)(crow(0, 1, crow_mode) - crow(0, 0, crow_mode)) * Np;
    const double* __restrict__ iK0 = invK + (long)a * Np * Np;
    const double* __restrict__ iK1 = iK0 + cstep;
    const double* __restrict__ iK2 = iK1 + cstep;
    const double* __restrict__ iK3 = iK2 + cstep;
    __shared__ double red[4];
    __shared__ double Cs[2][EMK + 2][64];
    __shared__ double Et[TAB == 1 ? EXPT_N : (TAB == 2 ? EXPT32_N : 1)];
    if (TAB == 1) exp_tab_fill(Et, etab, tid, 256);              // (visible behind the barrier that follows the first stage())
    if (TAB == 2) exp_tab32_fill(Et, tid);
    const int fr = lane & 15, fk = lane >> 4;
    double af[KD / 4], la[4], bai[4], racc[4];
    unsigned ko = 0;                                             // element offset of (row of register 0, current sub-tile's column)
    auto load_rows = [&](int tis, int jts) {
        const int i0 = tis * 64 + 16 * wave;
#pragma unroll
        for (int s4 = 0; s4 < KD / 4; ++s4) af[s4] = o[(long)(4 * s4 + fk) * Np + i0 + fr];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int irow = i0 + crow(lane, r, crow_mode);
            la[r] = La[irow];                                    // (padded rows: EM_PAD_LOG, their Q is an exact zero)
            bai[r] = irow < N ? ba[irow] : 0.0;
        }
        ko = (unsigned)(i0 + crow(lane, 0, crow_mode)) * (unsigned)Np + (unsigned)(jts * 64 + fr);
    };
    double st[NQ];
    auto fetch = [&](int jtf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid + 256 * q, rw = e >> 6, cl = e & 63, j = jtf * 64 + cl;
            double v = 0.0;
            if (rw < EMK) v = Wt[(long)rw * Np + j];
            else if (rw == EMK) v = Lb[j];
            else if (rw == EMK + 1) v = (j < N) ? ba[j] : 0.0;
            st[q] = v;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid + 256 * q, rw = e >> 6, cl = e & 63;
            if (rw < EMK + 2) Cs[buf][rw][cl] = st[q];
        }
    };
    load_rows(ti, jt);
    fetch(jt);
    stage(0);
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) racc[r] = 0.0;
    double kacc = 0.0, tot = 0.0;
    for (long n = lo; n < hi; ++n) {
        const bool last = n + 1 == hi, diag = jt == ti;
        const int jn = diag ? 0 : jt + 1;                          // the next tile's column (a new strip starts at 0)
        if (diag) {                                                // everything so far in this strip lies left of the diagonal
#pragma unroll
            for (int r = 0; r < 4; ++r) racc[r] *= 2.0;
            kacc *= 2.0;
        }
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            // K^-1 of this sub-tile: requested here, used by the last instruction of the sub-tile (behind the matrix products
            // and the exps), no second set of registers, no mask (Q is an exact zero in padded rows, K^-1 in padded columns)
            const double ik0 = iK0[ko], ik1 = iK1[ko], ik2 = iK2[ko], ik3 = iK3[ko];
            ko += 16;
            if (t == 0 && !last) fetch(jn);
            const int cl = 16 * t + fr;
            const double lbj = Cs[cur][EMK][cl];
            const double bj = Cs[cur][EMK + 1][cl];
            d4 c;
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = la[r] + lbj;
#pragma unroll
            for (int s4 = 0; s4 < KD / 4; ++s4) c = mfma16(af[s4], Cs[cur][4 * s4 + fk][cl], c);
            double q[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                q[r] = TAB == 2 ? exp_tab32(c[r], Et) : TAB == 1 ? exp_tab<CLAMP>(c[r], Et) : exp_lean(c[r]);
                racc[r] = fma(bj, q[r], racc[r]);
            }
            kacc = fma(ik0, q[0], kacc);
            kacc = fma(ik1, q[1], kacc);
            kacc = fma(ik2, q[2], kacc);
            kacc = fma(ik3, q[3], kacc);
        }
        if (diag) {                                                // the strip is complete
            tot += ((bai[0] * racc[0] + bai[1] * racc[1]) + (bai[2] * racc[2] + bai[3] * racc[3])) - kacc;
#pragma unroll
            for (int r = 0; r < 4; ++r) racc[r] = 0.0;
            kacc = 0.0;
            if (!last) load_rows(++ti, 0);
        }
        jt = jn;
        if (!last) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // a range that ends inside a strip: all of that part lies left of the diagonal
    tot += 2.0 * (((bai[0] * racc[0] + bai[1] * racc[1]) + (bai[2] * racc[2] + bai[3] * racc[3])) - kacc);
    const double acc = wave_sum(tot);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) *pout = (red[0] + red[1]) + (red[2] + red[3]);
}

// cov_ab = t_p * sum_slots partial;  cov_aa += sf_a^2;  cov -= mean mean^T;  symmetric fill.
// grid (B * P), 64 threads: one wave per (input, pair) -- lane l adds the slots l, l + 64, ..., the lanes meet in wave_sum's
// fixed butterfly.  (r01-r05: one THREAD per (input, pair) walked its 256 partial sums, a chain of dependent loads: 50-100 us
// per input at C3, 1.5-3 % of that step.)
// mpart (optional): the mean's chunk sums (em_mean_kernel) -- the pair's two means are added here in em_mean_finish_kernel's
// order and the a == b workgroup writes mean_a, so that no one-workgroup launch sits between the mean and the pair sums
// (on the a == b launch's queue such a launch waited 110 us for a place while the other launch's workgroups poured in).
__global__ void __launch_bounds__(64) em_finish_kernel(const double* __restrict__ partial, const double* __restrict__ prep,
                                                       const double* __restrict__ hyper, double* __restrict__ mean,
                                                       double* __restrict__ cov, int B, int Ny, int d, int nslots,
                                                       const double* __restrict__ mpart) {
    const int P = Ny * (Ny + 1) / 2, stride = d * d + 1;
    const long gid = blockIdx.x;
    const int b = (int)(gid / P), p = (int)(gid % P), lane = threadIdx.x;
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= p) ++a;
    const int bb = p - a * (a + 1) / 2;
    const double* __restrict__ ps = partial + ((long)b * P + p) * nslots;
    double s = 0.0;
    for (int k = lane; k < nslots; k += 64) s += ps[k];
    s = wave_sum(s);
    if (lane != 0) return;
    const double t = prep[((long)b * (Ny + P) + Ny + p) * stride + d * d];
    double v = t * s;
    if (a == bb) v += hyper[(long)a * (d + 2) + d] * hyper[(long)a * (d + 2) + d];
    double ma, mb;
    if (mpart) {
        ma = mb = 0.0;
        for (int ch = 0; ch < EM_MEAN_CHUNKS; ++ch) {
            ma += mpart[((long)b * Ny + a) * EM_MEAN_CHUNKS + ch];
            mb += mpart[((long)b * Ny + bb) * EM_MEAN_CHUNKS + ch];
        }
        if (a == bb) mean[(long)b * Ny + a] = ma;
    } else {
        ma = mean[(long)b * Ny + a];
        mb = mean[(long)b * Ny + bb];
    }
    v -= ma * mb;
    cov[((long)b * Ny + a) * Ny + bb] = v;
    cov[((long)b * Ny + bb) * Ny + a] = v;
}


// ---- derivative outputs of the exact moments ---------------------------------------------------
// d mean / d(mu, Sigma) and d cov / d(mu, Sigma) of gp_exact_moment (what CasADi's AD hands to IPOPT when 'EM' is the
// MPC's propagation method, gp_class.py:220-224).  With W = A o Q of an ORDERED output pair (a, c) (rows belong to
// a, columns to c; W^(c,a) = W^(a,c)^T), v_i = x_i - mu, ii_i = v_i / ell_a^2, the kernel leaves per 64-COLUMN strip
//     s0 = sum_j c_j,   C1 = sum_j c_j v_j,   C2 = sum_j c_j v_j v_j^T,   X' = sum_j h_j v_j^T,
//     c_j = sum_i W_ij (column sums),   h_j = sum_i W_ij ii_i,
// from which em_sens_finish_kernel assembles, for the unordered pair (a >= c), with Lab = 1/ell_a^2 + 1/ell_c^2,
// G = (Lab Sigma + I)^-1 (row sums of (a,c) are column sums of (c,a), so the kernel runs over ordered pairs):
//     z1 = C1^(c,a)/ell_a^2 + C1^(a,c)/ell_c^2
//     ZZ = L_a^-1 C2^(c,a) L_a^-1 + L_c^-1 C2^(a,c) L_c^-1 + 2 X'^(a,c) L_c^-1        (maha's cross term is not symmetrised)
//     d(t s)/d mu = t G z1,    d(t s)/d Sigma = t (-1/2 G Lab s0 + 1/2 G ZZ G^T).
// The column moments are a matrix product: a 16 x 16 tile of W sits in the lanes in the accumulator layout, which IS the
// A-operand layout of its transpose, so  [c_j | h_j] += W^T [1 | ii]  costs four v_mfma_f64_16x16x4_f64 per tile on the
// matrix pipe, which the value kernel leaves two thirds idle, instead of nine VALU fma per entry (the first version: 64
// accumulator registers per lane, one wave per SIMD, 8.2 ms per input at C3 against 1.5 ms for the value).
// (all of it for a cross-term depth KD = 8 or 16 = em_depth(d), as the value kernels)
constexpr int em_nss(int KD) { return 1 + KD + 2 * KD * KD; }     // values per (input, ordered pair, strip)
// operand rows per (input, ordered pair): row side [U (KD) | La | beta_a], column side [Wt (KD) | Lb | beta_c], ii (KD)
constexpr int em_ops_ord(int KD) { return 3 * KD + 4; }

// operands for ORDERED pairs, pair index po = a * Ny + c.  beta rows are copied in (zero in padded points) so that the
// pair kernel's tile fetch is one branch-free block of rows.
template <int KD>
__global__ void __launch_bounds__(256) em_operands_ordered_kernel(const double* __restrict__ XT, const double* __restrict__ Z,
                                                                  const double* __restrict__ hyper,
                                                                  const double* __restrict__ prep,
                                                                  const double* __restrict__ beta, double* __restrict__ ops,
                                                                  int N, int Np, int d, int Ny, int b0) {
    constexpr int EMK = KD, EM_OPS_ORD = em_ops_ord(KD), EM_ROW0 = 0, EM_COL0 = KD + 2, EM_II0 = 2 * KD + 4;
    const int P = Ny * (Ny + 1) / 2, stride = d * d + 1;
    const int i = blockIdx.x * 256 + threadIdx.x, po = blockIdx.y, bl = blockIdx.z, b = b0 + bl;
    if (i >= Np) return;
    const int a = po / Ny, c = po % Ny, hi = a > c ? a : c, lo = a > c ? c : a;
    const double* S = prep + ((long)b * (Ny + P) + Ny + hi * (hi + 1) / 2 + lo) * stride;
    const double* ha = hyper + (long)a * (d + 2);
    const double* hb = hyper + (long)c * (d + 2);
    double* o = ops + ((long)bl * Ny * Ny + po) * EM_OPS_ORD * Np;
    double v[DMAX], ii[DMAX], ij[DMAX];
    double lka = 0.0, lkb = 0.0;
    for (int k = 0; k < d; ++k) {
        v[k] = (i < N) ? XT[(long)k * Np + i] - Z[(long)b * d + k] : 0.0;
        ii[k] = v[k] / (ha[k] * ha[k]);
        ij[k] = v[k] / (hb[k] * hb[k]);
        lka += v[k] * v[k] / (ha[k] * ha[k]);
        lkb += v[k] * v[k] / (hb[k] * hb[k]);
    }
    double qa = 0.0, qb = 0.0;
    for (int cc = 0; cc < EMK; ++cc) {
        double ua = 0.0, ub = 0.0;
        if (cc < d)
            for (int k = 0; k < d; ++k) { ua += ii[k] * S[k * d + cc]; ub += ij[k] * S[k * d + cc]; }
        o[(long)(EM_ROW0 + cc) * Np + i] = 2.0 * ua;
        o[(long)(EM_COL0 + cc) * Np + i] = (cc < d) ? ij[cc] : 0.0;
        o[(long)(EM_II0 + cc) * Np + i] = (cc < d) ? ii[cc] : 0.0;
        if (cc < d) { qa += ua * ii[cc]; qb += ub * ij[cc]; }
    }
    o[(long)(EM_ROW0 + EMK) * Np + i] = (2.0 * log(ha[d]) - 0.5 * lka) + qa;
    o[(long)(EM_ROW0 + EMK + 1) * Np + i] = (i < N) ? beta[(long)a * Np + i] : 0.0;
    o[(long)(EM_COL0 + EMK) * Np + i] = (2.0 * log(hb[d]) - 0.5 * lkb) + qb;
    o[(long)(EM_COL0 + EMK + 1) * Np + i] = (i < N) ? beta[(long)c * Np + i] : 0.0;
}

// grid (Np/64 column strips, Ny*Ny, Bc), 256 threads: wave w owns columns 64 tj + 16 w .. + 15 and sweeps the rows in
// 64-row tiles staged through LDS (one-tile prefetch).  A lane's four accumulator rows of a 16-row sub-tile sit side
// by side in LDS (position pos(il) below), so that La, beta_a and the feature operand of the moment products come
// in as 128-bit reads.  part[((bl*Ny*Ny + po)*tiles + strip)*EM_NSS + e].  Launched once per kind like
// em_pair_kernel: only the a == c variant carries the K^-1 registers.
template <bool DIAG, int KD>
__global__ void __launch_bounds__(256) em_pair_sens_kernel(const double* __restrict__ ops, const double* __restrict__ invK,
                                                           const double* __restrict__ XT, const double* __restrict__ Z,
                                                           double* __restrict__ part, int N, int Np, int Ny, int d, int b0,
                                                           int crow_mode) {
    constexpr int EMK = KD, EM_OPS_ORD = em_ops_ord(KD), EM_NSS = em_nss(KD), EM_ROW0 = 0, EM_COL0 = KD + 2, EM_II0 = 2 * KD + 4;
    constexpr int NF = KD == 8 ? 1 : 0;          // KD = 8: feature column 0 is the constant 1 (c_j rides the moment product);
                                                 // KD = 16: all 16 feature columns are ii, c_j is summed on the VALU
    const int tj = blockIdx.x, po = blockIdx.y, bl = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = Np / 64, a = po / Ny, cb = po % Ny;
    if ((a == cb) != DIAG) return;
    const double* __restrict__ o = ops + ((long)bl * Ny * Ny + po) * EM_OPS_ORD * Np;
    const double* __restrict__ iK = invK + (long)a * Np * Np;
    __shared__ double Us[2][EMK][64];             // row tile: U (A fragments of the cross term)
    __shared__ __attribute__((aligned(32))) double LBs[2][64][2];              // row tile: {La, beta_a} at pos(il)
    __shared__ __attribute__((aligned(32))) double Fs[2][16][16][4];           // row tile: features [sub-tile*4 + lane group][f][r];  f = 0: 1, 1..8: ii, 9..15: 0
    __shared__ double Ds[64][EMK + 1];            // per strip column: c_j, h_j[EMK]
    __shared__ double Vs[64][EMK];                // per strip column: v_j
    const int fr = lane & 15, fk = lane >> 4, n0 = tj * 64, col = n0 + 16 * wave + fr;
    const double* __restrict__ oc = o + (long)EM_COL0 * Np;
    double bf[KD / 4];                                                                  // B fragments: constant over the sweep
#pragma unroll
    for (int s4 = 0; s4 < KD / 4; ++s4) bf[s4] = oc[(long)(4 * s4 + fk) * Np + col];
    const double lbj = oc[(long)EMK * Np + col], bj = oc[(long)(EMK + 1) * Np + col];
    for (int e = tid; e < 64 * EMK; e += 256) {
        const int cl = e / EMK, k = e % EMK, j = n0 + cl;
        Vs[cl][k] = (k < d && j < N) ? XT[(long)k * Np + j] - Z[(long)(b0 + bl) * d + k] : 0.0;
    }
    for (int e = tid; e < 2 * 16 * 16 * 4; e += 256) {     // constant feature columns of both buffers
        const int f = (e >> 2) & 15;
        if (f < NF || f >= NF + EMK) (&Fs[0][0][0][0])[e] = f < NF ? 1.0 : 0.0;
    }
    // position of row il (0..63) of a tile: sub-tile sb = il / 16, then (lane group g, register r) with crow(g*16, r) == il % 16
    auto pos = [&](int il) {
        const int w16 = il & 15, g = crow_mode == 0 ? (w16 & 3) : (w16 >> 2), r = crow_mode == 0 ? (w16 >> 2) : (w16 & 3);
        return ((il >> 4) * 4 + g) * 4 + r;
    };
    const int scl = tid & 63, srw = tid >> 6, sp = pos(scl), spg = sp >> 2, spr = sp & 3;
    constexpr int NG = KD / 4;           // row groups of four: thread (srw, scl) handles rows 4 g + srw of U and of ii
    double stU[NG], stI[NG], stL;
    auto fetch = [&](int it) {           // rows U (KD), La, beta_a and the KD ii rows of 64 points: 2 KD / 4 + 1 coalesced loads per thread
        const long i = (long)it * 64 + scl;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            stU[g] = o[(long)(EM_ROW0 + 4 * g + srw) * Np + i];
            stI[g] = o[(long)(EM_II0 + 4 * g + srw) * Np + i];
        }
        stL = o[(long)(EM_ROW0 + EMK + (srw & 1)) * Np + i];
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            Us[buf][4 * g + srw][scl] = stU[g];
            Fs[buf][spg][NF + 4 * g + srw][spr] = stI[g];
        }
        if (srw < 2) LBs[buf][sp][srw] = stL;
    };
    d4 D0 = d4{0.0, 0.0, 0.0, 0.0}, D1 = D0;   // [c_j | h_j] of this wave's 16 columns: D[j = crow(lane, r)][f = lane & 15]
    double csum = 0.0;                         // KD = 16: this lane's share of c_j, j = column lane & 15 (rows of its registers)
    fetch(0);
    stage(0);
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < tiles; ++it) {
        if (it + 1 < tiles) fetch(it + 1);
        double ik[4][4];
        if (DIAG) {
#pragma unroll
            for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = it * 64 + 16 * sb + crow(lane, r, crow_mode);
                    ik[sb][r] = i < N ? iK[(long)i * Np + col] : 0.0;
                }
        }
        d4 c[4];
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {                                 // cross terms of the four 16-row sub-tiles
            c[sb] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < KD / 4; ++s4) c[sb] = mfma16(Us[cur][4 * s4 + fk][16 * sb + fr], bf[s4], c[sb]);
        }
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            const double4 lb01 = *reinterpret_cast<const double4*>(&LBs[cur][(sb * 4 + fk) * 4][0]);
            const double4 lb23 = *reinterpret_cast<const double4*>(&LBs[cur][(sb * 4 + fk) * 4 + 2][0]);
            const double la[4] = {lb01.x, lb01.z, lb23.x, lb23.z}, bai[4] = {lb01.y, lb01.w, lb23.y, lb23.w};
            const double4 ft = *reinterpret_cast<const double4*>(&Fs[cur][sb * 4 + fk][fr][0]);
            const double f4[4] = {ft.x, ft.y, ft.z, ft.w};
            double w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double q = exp_lean((la[r] + lbj) + c[sb][r]);
                double wgt = bai[r] * bj;
                if (DIAG) wgt -= ik[sb][r];
                w[r] = wgt * q;
            }
            // D[j][f] += sum over the 4 rows register r holds: the accumulator layout of W is the A layout of W^T
            D0 = mfma16(w[0], f4[0], D0);
            D1 = mfma16(w[1], f4[1], D1);
            D0 = mfma16(w[2], f4[2], D0);
            D1 = mfma16(w[3], f4[3], D1);
            if (NF == 0) csum += (w[0] + w[1]) + (w[2] + w[3]);
        }
        if (it + 1 < tiles) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // Ds[column][0] = c_j, Ds[column][1 + k] = h_j[k]
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (fr < NF + EMK) Ds[16 * wave + crow(lane, r, crow_mode)][fr + (1 - NF)] = D0[r] + D1[r];
    if (NF == 0) {             // the rows of a column are spread over the four lane groups: fixed-order sum across them
        csum += __shfl_xor(csum, 16);
        csum += __shfl_xor(csum, 32);
        if (lane < 16) Ds[16 * wave + lane][0] = csum;
    }
    __syncthreads();
    for (int t = tid; t < EM_NSS; t += 256) {        // fixed-order sums over the strip's 64 columns
        double s = 0.0;
        if (t == 0) {
            for (int cl = 0; cl < 64; ++cl) s += Ds[cl][0];
        } else if (t < 1 + EMK) {
            const int k = t - 1;
            for (int cl = 0; cl < 64; ++cl) s += Ds[cl][0] * Vs[cl][k];
        } else if (t < 1 + EMK + EMK * EMK) {
            const int e = t - 1 - EMK, k = e / EMK, l = e % EMK;
            for (int cl = 0; cl < 64; ++cl) s += Ds[cl][0] * Vs[cl][k] * Vs[cl][l];
        } else {
            const int e = t - 1 - EMK - EMK * EMK, k = e / EMK, l = e % EMK;
            for (int cl = 0; cl < 64; ++cl) s += Ds[cl][1 + k] * Vs[cl][l];
        }
        part[(((long)bl * Ny * Ny + po) * tiles + tj) * EM_NSS + t] = s;
    }
}

// sums[(bl*Ny*Ny + po)*EM_NSS + e] = sum over strips (fixed order).  grid (Ny*Ny, Bc), 256 threads.
template <int KD>
__global__ void __launch_bounds__(256) em_sens_reduce_kernel(const double* __restrict__ part, double* __restrict__ sums,
                                                             int Ny, int tiles) {
    constexpr int EM_NSS = em_nss(KD);
    const int po = blockIdx.x, bl = blockIdx.y;
    for (int e = threadIdx.x; e < EM_NSS; e += 256) {
        const double* p = part + (((long)bl * Ny * Ny + po) * tiles) * EM_NSS + e;
        double s = 0.0;
        for (int t = 0; t < tiles; ++t) s += p[(long)t * EM_NSS];
        sums[((long)bl * Ny * Ny + po) * EM_NSS + e] = s;
    }
}

// d mean_a / d mu = P_a M1, d mean_a / d Sigma = -1/2 P_a mean_a + 1/2 P_a M2 P_a with M1 = sum_i w_i v_i,
// M2 = sum_i w_i v_i v_i^T, w_i = beta_ai q_ai, P_a = (Sigma + Lambda_a)^-1 (prep's iR).  grid (Ny, Bc), 256 threads.
template <int KD>
__global__ void __launch_bounds__(256) em_mean_sens_kernel(const double* __restrict__ XT, const double* __restrict__ Z,
                                                           const double* __restrict__ beta, const double* __restrict__ prep,
                                                           double* __restrict__ dm_dz, double* __restrict__ dm_dS, int N,
                                                           int Np, int d, int Ny, int b0) {
    constexpr int EMK = KD, NM = 1 + EMK + EMK * (EMK + 1) / 2;
    // The moments M0, M1[k], M2[k][l <= k] are gathered in passes of KC rows k (KD = 8: one pass of 45 accumulators per
    // thread as before; KD = 16: eight passes of 2 + 32 -- 153 at once would not fit the register file, 4 + 64 still spilled); every
    // pass recomputes the weights w_i, KD^2 flops per point against N x KD x 8 bytes of coordinates: nothing.
    constexpr int KC = KD == 8 ? 8 : 2;
    const int a = blockIdx.x, b = b0 + blockIdx.y, tid = threadIdx.x;
    const int P = Ny * (Ny + 1) / 2, stride = d * d + 1;
    const double* pr = prep + ((long)b * (Ny + P) + a) * stride;
    __shared__ double iR[EMK * EMK], mu[EMK], red[4][NM], M[NM], T1[EMK * EMK];
    for (int e = tid; e < EMK * EMK; e += 256) iR[e] = (e / EMK < d && e % EMK < d) ? pr[(e / EMK) * d + e % EMK] : 0.0;
    if (tid < EMK) mu[tid] = tid < d ? Z[(long)b * d + tid] : 0.0;
    __syncthreads();
    const double c = pr[d * d];
#pragma unroll 1
    for (int k0 = 0; k0 < EMK; k0 += KC) {
        double a0 = 0.0, a1[KC], a2[KC][EMK];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            a1[kk] = 0.0;
#pragma unroll
            for (int l = 0; l < EMK; ++l) a2[kk][l] = 0.0;
        }
        for (int i = tid; i < N; i += 256) {
            double v[EMK];
#pragma unroll
            for (int k = 0; k < EMK; ++k) v[k] = k < d ? XT[(long)k * Np + i] - mu[k] : 0.0;
            // qf = v^T iR v = sum_k v_k (iR v)_k (iR is symmetric): the row index k stays a loop variable at KD = 16 -- fully
            // unrolled, the 256 LDS operands were hoisted into registers and the kernel spilled 290 of them
            double qf = 0.0;
#pragma unroll(KD == 8 ? 8 : 1)
            for (int k = 0; k < EMK; ++k) {
                double t = 0.0;
#pragma unroll
                for (int r = 0; r < EMK; ++r) t += iR[k * EMK + r] * v[r];
                double vk = 0.0;
#pragma unroll
                for (int q = 0; q < EMK; ++q) vk = (q == k) ? v[q] : vk;
                qf += t * vk;
            }
            const double w = c * exp(-0.5 * qf) * beta[(long)a * Np + i];
            a0 += w;
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                double vk = 0.0;                       // v[k0 + kk] without a run-time register index
#pragma unroll
                for (int q = 0; q < EMK; ++q) vk = (q == k0 + kk) ? v[q] : vk;
                a1[kk] += w * vk;
#pragma unroll
                for (int l = 0; l < EMK; ++l) a2[kk][l] += w * vk * v[l];      // (entries l > k are formed and not used)
            }
        }
        // wave sums -> red, for the entries of this pass
        {
            const double t = wave_sum(a0);
            if ((tid & 63) == 0 && k0 == 0) red[tid >> 6][0] = t;
        }
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const int k = k0 + kk;
            const double t = wave_sum(a1[kk]);
            if ((tid & 63) == 0) red[tid >> 6][1 + k] = t;
#pragma unroll
            for (int l = 0; l < EMK; ++l) {
                const double t2 = wave_sum(a2[kk][l]);
                if ((tid & 63) == 0 && l <= k) red[tid >> 6][1 + EMK + k * (k + 1) / 2 + l] = t2;
            }
        }
    }
    __syncthreads();
    if (tid < NM) M[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    __syncthreads();
    auto m2 = [&](int k, int l) { const int hi = k > l ? k : l, lo = k > l ? l : k; return M[1 + EMK + hi * (hi + 1) / 2 + lo]; };
    if (tid < d) {                                   // d mean / d mu = P M1
        double s = 0.0;
        for (int k = 0; k < d; ++k) s += iR[tid * EMK + k] * M[1 + k];
        dm_dz[((long)blockIdx.y * Ny + a) * d + tid] = s;
    }
    if (tid < d * d) {                               // T1 = P M2
        const int r = tid / d, q = tid % d;
        double s = 0.0;
        for (int k = 0; k < d; ++k) s += iR[r * EMK + k] * m2(k, q);
        T1[r * EMK + q] = s;
    }
    __syncthreads();
    if (tid < d * d) {
        const int r = tid / d, q = tid % d;
        double s = 0.0;
        for (int k = 0; k < d; ++k) s += T1[r * EMK + k] * iR[k * EMK + q];
        dm_dS[(((long)blockIdx.y * Ny + a) * d + r) * d + q] = -0.5 * iR[r * EMK + q] * M[0] + 0.5 * s;
    }
}

// assembly per (input, unordered pair a >= c).  grid (Bc * P), DMAX * GJ_LD threads: one workgroup per item, its d x d
// algebra in LDS with a thread per entry (one THREAD per item with the matrices in scratch memory took 0.31 ms at C3).
template <int KD>
__global__ void __launch_bounds__(DMAX * GJ_LD) em_sens_finish_kernel(const double* __restrict__ sums, const double* __restrict__ prep,
                                                                      const double* __restrict__ hyper, const double* __restrict__ Sigma,
                                                                      const double* __restrict__ mean, const double* __restrict__ dm_dz,
                                                                      const double* __restrict__ dm_dS, double* __restrict__ dc_dz,
                                                                      double* __restrict__ dc_dS, int Bc, int Ny, int d, int b0) {
    constexpr int EMK = KD, EM_NSS = em_nss(KD);
    const int P = Ny * (Ny + 1) / 2, stride = d * d + 1;
    const int tid = threadIdx.x, r = tid / GJ_LD, q = tid % GJ_LD;
    const int bl = (int)blockIdx.x / P, p = (int)blockIdx.x % P, b = b0 + bl;
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= p) ++a;
    const int c = p - a * (a + 1) / 2;
    const double* ha = hyper + (long)a * (d + 2);
    const double* hc = hyper + (long)c * (d + 2);
    const double* Sg = Sigma + (long)b * d * d;
    const double t = prep[((long)b * (Ny + P) + Ny + p) * stride + d * d];
    const double* sa_ = sums + ((long)bl * Ny * Ny + a * Ny + c) * EM_NSS;      // ordered (a, c)
    const double* sc_ = sums + ((long)bl * Ny * Ny + c * Ny + a) * EM_NSS;      // ordered (c, a)
    __shared__ double M[DMAX * GJ_LD], ila[EMK], ilc[EMK], lab[EMK], z1[EMK], ZZ[EMK * EMK], T1[EMK * EMK];
    __shared__ int piv[2];
    if (tid < d) {
        ila[tid] = 1.0 / (ha[tid] * ha[tid]);
        ilc[tid] = 1.0 / (hc[tid] * hc[tid]);
        lab[tid] = ila[tid] + ilc[tid];
    }
    __syncthreads();
    // G = (Lab Sigma + I)^-1: the right half of [Lab Sigma + I | I] after the elimination
    if (r < d && q < 2 * d)
        M[r * GJ_LD + q] = q < d ? lab[r] * Sg[r * d + q] + (r == q ? 1.0 : 0.0) : (q - d == r ? 1.0 : 0.0);
    __syncthreads();
    gauss_jordan_lds(M, d, 2 * d, piv);
    auto G = [&](int k, int l) { return M[k * GJ_LD + d + l]; };
    const double s0 = sa_[0];
    // column moments of ordered (c, a) are the row moments of (a, c)
    if (tid < d) z1[tid] = ila[tid] * sc_[1 + tid] + ilc[tid] * sa_[1 + tid];
    if (r < d && q < d) {
        const double* C2r = sc_ + 1 + EMK;
        const double* C2c = sa_ + 1 + EMK;
        const double* Xp = sa_ + 1 + EMK + EMK * EMK;
        ZZ[r * EMK + q] = ila[r] * C2r[r * EMK + q] * ila[q] + ilc[r] * C2c[r * EMK + q] * ilc[q] + 2.0 * Xp[r * EMK + q] * ilc[q];
    }
    __syncthreads();
    const double ma = mean[(long)b * Ny + a], mc = mean[(long)b * Ny + c];
    const double* dza = dm_dz + ((long)bl * Ny + a) * d;
    const double* dzc = dm_dz + ((long)bl * Ny + c) * d;
    const double* dSa = dm_dS + ((long)bl * Ny + a) * d * d;
    const double* dSc = dm_dS + ((long)bl * Ny + c) * d * d;
    if (tid < d) {
        double s = 0.0;
        for (int l = 0; l < d; ++l) s += G(tid, l) * z1[l];
        const double v = t * s - mc * dza[tid] - ma * dzc[tid];
        dc_dz[(((long)bl * Ny + a) * Ny + c) * d + tid] = v;
        dc_dz[(((long)bl * Ny + c) * Ny + a) * d + tid] = v;
    }
    if (r < d && q < d) {
        double s = 0.0;
        for (int m = 0; m < d; ++m) s += G(r, m) * ZZ[m * EMK + q];
        T1[r * EMK + q] = s;
    }
    __syncthreads();
    if (r < d && q < d) {
        double s = 0.0;
        for (int m = 0; m < d; ++m) s += T1[r * EMK + m] * G(q, m);       // (G ZZ G^T)_rq
        const double v = t * (-0.5 * G(r, q) * lab[q] * s0 + 0.5 * s) - mc * dSa[r * d + q] - ma * dSc[r * d + q];
        dc_dS[((((long)bl * Ny + a) * Ny + c) * d + r) * d + q] = v;
        dc_dS[((((long)bl * Ny + c) * Ny + a) * d + r) * d + q] = v;
    }
}

// ---- legacy methods a12 ------------------------------------------------------------------------------
// 'old_ME' (gp, gp_functions.py:176-256) and 'old_TA' (gp_taylor_approx(diag=True), :259-340) both start
// from u = K_a^-1 ks (one GEMM for the whole batch: UT = KsT K^-1).  This kernel turns (ks, u) into the
// scalars those functions need.  grid (B, Ny), 256 threads.  out[(b*Ny + a)*4 + {0,1,2,3}] =
//   mean = u . y (:237,246 with alpha=None),  var = kss - u . ks (:249),
//   p0   = sum_i v_i0 ks_i u_i,  dm = w_a[a] sum_i v_ia ks_i beta_ai  (d_mean, :322).
__global__ void __launch_bounds__(256) legacy_scalars_kernel(const double* __restrict__ XT, const double* __restrict__ Z,
                                                             const double* __restrict__ hyper, const double* __restrict__ Y,
                                                             const double* __restrict__ beta, const double* __restrict__ KsT,
                                                             const double* __restrict__ UT, double* __restrict__ out,
                                                             int N, int Np, int d, int Bp, int Ny) {
    const int b = blockIdx.x, a = blockIdx.y, tid = threadIdx.x;
    __shared__ double red[4][4];
    const double* ks = KsT + ((long)a * Bp + b) * Np;
    const double* u = UT + ((long)a * Bp + b) * Np;
    const double* hy = hyper + (long)a * (d + 2);
    const double z0 = Z[(long)b * d], za = Z[(long)b * d + a];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int i = tid; i < N; i += 256) {
        const double k = ks[i], ui = u[i];
        s0 += ui * Y[(long)a * Np + i];
        s1 += ui * k;
        s2 += (XT[i] - z0) * k * ui;
        s3 += (XT[(long)a * Np + i] - za) * k * beta[(long)a * Np + i];
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
    if ((tid & 63) == 0) { red[tid >> 6][0] = s0; red[tid >> 6][1] = s1; red[tid >> 6][2] = s2; red[tid >> 6][3] = s3; }
    __syncthreads();
    if (tid < 4) {
        double s = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        if (tid == 1) s = hy[d] * hy[d] - s;
        if (tid == 3) s = s / (hy[a] * hy[a]);
        out[((long)b * Ny + a) * 4 + tid] = s;
    }
}

// Assemble the legacy outputs.  One thread per input b.
//   old_ME: cov = diag(var).
//   old_TA (gp_functions.py:303-338, restated literally incl. its self-documented bug :325): only entry
//   [0,0] of covar_temp is non-zero, so
//     cov[a,a] = var_a + Sigma[a,a] (0.5 dd_var[0,0] + d_mean[0]^2),
//     dd_var[0,0] = -2 w_a0^2 (v_00 p0_a + v_00^2 (ks_a . u_a)) + 2 w_a0 (sf_a^2 - var_0),
//   with v_00 = X[0,0] - z_0 (CasADi linear indexing v[e], e = 0) and var_0 / d_mean[0] from output 0.
__global__ void __launch_bounds__(64) legacy_finish_kernel(const double* __restrict__ sc, const double* __restrict__ XT,
                                                           const double* __restrict__ Z, const double* __restrict__ hyper,
                                                           const double* __restrict__ Sigma, double* __restrict__ mean,
                                                           double* __restrict__ cov, int B, int Ny, int d, int old_ta) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const double v00 = XT[0] - Z[(long)b * d];
    const double var0 = sc[((long)b * Ny) * 4 + 1], dm0 = sc[((long)b * Ny) * 4 + 3];
    for (int a = 0; a < Ny; ++a) {
        const double* s = sc + ((long)b * Ny + a) * 4;
        const double* hy = hyper + (long)a * (d + 2);
        mean[(long)b * Ny + a] = s[0];
        double c = s[1];
        if (old_ta) {
            const double w0 = 1.0 / (hy[0] * hy[0]);
            const double ksu = hy[d] * hy[d] - s[1];
            const double dd00 = -2.0 * w0 * w0 * (v00 * s[2] + v00 * v00 * ksu) + 2.0 * w0 * (hy[d] * hy[d] - var0);
            c = s[1] + Sigma[(long)b * d * d + a * d + a] * (0.5 * dd00 + dm0 * dm0);
        }
        for (int e = 0; e < Ny; ++e) cov[((long)b * Ny + a) * Ny + e] = (e == a) ? c : 0.0;
    }
}

}  // namespace gpmpc
