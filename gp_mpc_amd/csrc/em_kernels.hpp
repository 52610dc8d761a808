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

// per-input small algebra.  Layout of `prep` per input b (doubles):
//   [a < Ny]       iR_a[d*d], c_a                              -> Ny * (d*d + 1)
//   [pair p]       S_p[d*d], t_p                               -> P  * (d*d + 1),  p = a(a+1)/2 + b
// grid (ceil(B*(Ny+P)/64)), 64 threads: one thread per (input, item).
__global__ void __launch_bounds__(64) em_prep_kernel(const double* __restrict__ hyper, const double* __restrict__ Sigma,
                                                     double* __restrict__ prep, int B, int Ny, int d) {
    const int P = Ny * (Ny + 1) / 2, items = Ny + P;
    const long gid = (long)blockIdx.x * 64 + threadIdx.x;
    if (gid >= (long)B * items) return;
    const int b = (int)(gid / items), it = (int)(gid % items);
    const double* Sg = Sigma + (long)b * d * d;
    const int stride = d * d + 1;
    double* out = prep + ((long)b * items + it) * stride;
    double A[DMAX * DMAX], R[DMAX * DMAX];
    if (it < Ny) {
        const double* hy = hyper + (long)it * (d + 2);
        // iR = iLambda (I - (I + Sigma iLambda)^-1 (Sigma iLambda)),  R = Sigma + Lambda
        for (int r = 0; r < d; ++r)
            for (int c = 0; c < d; ++c) {
                const double sil = Sg[r * d + c] / (hy[c] * hy[c]);
                A[r * d + c] = sil + (r == c ? 1.0 : 0.0);
                R[r * d + c] = sil;                 // right-hand side: Sigma iLambda
            }
        small_solve(A, R, d, d);                    // R <- (I + Sigma iLambda)^-1 Sigma iLambda
        for (int r = 0; r < d; ++r)
            for (int c = 0; c < d; ++c) out[r * d + c] = ((r == c ? 1.0 : 0.0) - R[r * d + c]) / (hy[r] * hy[r]);
        double prod = 1.0;
        for (int r = 0; r < d; ++r)
            for (int c = 0; c < d; ++c) A[r * d + c] = Sg[r * d + c] + (r == c ? hy[r] * hy[r] : 0.0);
        for (int r = 0; r < d; ++r) prod *= hy[r];
        double dummy[1];
        const double det = small_solve(A, dummy, d, 0);
        out[d * d] = hy[d] * hy[d] / sqrt(det) * prod;
    } else {
        int p = it - Ny, a = 0;
        while ((a + 1) * (a + 2) / 2 <= p) ++a;
        const int bb = p - a * (a + 1) / 2;
        const double* ha = hyper + (long)a * (d + 2);
        const double* hb = hyper + (long)bb * (d + 2);
        for (int r = 0; r < d; ++r)
            for (int c = 0; c < d; ++c) {
                A[r * d + c] = Sg[r * d + c] * (1.0 / (ha[c] * ha[c]) + 1.0 / (hb[c] * hb[c])) + (r == c ? 1.0 : 0.0);
                R[r * d + c] = Sg[r * d + c] * 0.5;
            }
        const double det = small_solve(A, R, d, d);  // R <- R2^-1 (Sigma / 2)
        for (int e = 0; e < d * d; ++e) out[e] = R[e];
        out[d * d] = 1.0 / sqrt(det);
    }
}

// mean_a = sum_i q_i beta_ai.  grid (Ny, B), 256 threads.
__global__ void __launch_bounds__(256) em_mean_kernel(const double* __restrict__ XT, const double* __restrict__ Z,
                                                      const double* __restrict__ beta, const double* __restrict__ prep,
                                                      double* __restrict__ mean, int N, int Np, int d, int Ny) {
    const int a = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int P = Ny * (Ny + 1) / 2, stride = d * d + 1;
    const double* pr = prep + ((long)b * (Ny + P) + a) * stride;
    __shared__ double iR[DMAX * DMAX], mu[DMAX], red[4];
    for (int e = tid; e < d * d; e += 256) iR[e] = pr[e];
    if (tid < d) mu[tid] = Z[(long)b * d + tid];
    __syncthreads();
    const double c = pr[d * d];
    double s = 0.0;
    for (int i = tid; i < N; i += 256) {
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
    if (tid == 0) mean[(long)b * Ny + a] = (red[0] + red[1]) + (red[2] + red[3]);
}

constexpr int EMK = 8;   // cross-term depth handled by the MFMA path (d <= 8: two 16x16x4 steps)

// Per-(input, pair, point) operands of the pair kernel.  One thread per (b, p, i); arrays are
// [b][p][...][Np] so that the pair kernel's loads are contiguous in the point index:
//   U[k][i]  = (ii_i S)_k * 2        (A operand of the cross-term product, k < EMK, zero padded)
//   Wt[k][j] = ij_j,k                 (B operand)
//   La[i] = log_k[i,a] + ii_i S ii_i^T,   Lb[j] = log_k[j,b] + ij_j S ij_j^T
// (gp_functions.py:394-396, :400-408 with maha expanded as in the header comment).
__global__ void __launch_bounds__(256) em_operands_kernel(const double* __restrict__ XT, const double* __restrict__ Z,
                                                          const double* __restrict__ hyper, const double* __restrict__ prep,
                                                          double* __restrict__ ops, int N, int Np, int d, int Ny) {
    const int P = Ny * (Ny + 1) / 2, stride = d * d + 1;
    const int i = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y, b = blockIdx.z;
    if (i >= Np) return;
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
    double qa = 0.0, qb = 0.0;
    for (int c = 0; c < EMK; ++c) {
        double ua = 0.0, ub = 0.0;
        if (c < d)
            for (int k = 0; k < d; ++k) { ua += ii[k] * S[k * d + c]; ub += ij[k] * S[k * d + c]; }
        o[(long)c * Np + i] = 2.0 * ua;
        o[(long)(EMK + c) * Np + i] = (c < d) ? ij[c] : 0.0;
        if (c < d) { qa += ua * ii[c]; qb += ub * ij[c]; }
    }
    o[(long)(2 * EMK) * Np + i] = (2.0 * log(ha[d]) - 0.5 * lka) + qa;
    o[(long)(2 * EMK + 1) * Np + i] = (2.0 * log(hb[d]) - 0.5 * lkb) + qb;
}

// Pair sums on a 64-row strip.  grid (Np/64, P, B), 256 threads = 4 waves; wave w owns rows 16w..16w+15 of
// the strip and sweeps the columns in 64-wide tiles without any barrier: the cross terms
// 2 (ii_i S) . ij_j of a 16 x 16 tile are two v_mfma_f64_16x16x4_f64 (depth EMK = 8, zero padded); the VALU
// adds La_i + Lb_j, takes the lean exp and accumulates (beta_ai beta_bj - [a==b] K^-1_ij) Q_ij.  For a == b
// the summand is symmetric in (i, j): only column tiles up to the diagonal are visited, off-diagonal
// tiles counted twice.  partial[(b*P + p)*tiles + strip].
template <bool DIAG>
__global__ void __launch_bounds__(256) em_pair_kernel(const double* __restrict__ ops, const double* __restrict__ beta,
                                                      const double* __restrict__ invK, double* __restrict__ partial,
                                                      int N, int Np, int Ny, int crow_mode) {
    const int ti = blockIdx.x, p = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = Ny * (Ny + 1) / 2, tiles = Np / 64;
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= p) ++a;
    const int bb = p - a * (a + 1) / 2;
    if ((a == bb) != DIAG) return;   // launched once per kind: the a == b variant carries the K^-1 registers
    constexpr bool diag = DIAG;
    const double* __restrict__ o = ops + ((long)b * P + p) * (2 * EMK + 2) * Np;
    const double* __restrict__ Wt = o + (long)EMK * Np;
    const double* __restrict__ La = o + (long)(2 * EMK) * Np;
    const double* __restrict__ Lb = o + (long)(2 * EMK + 1) * Np;
    const double* __restrict__ ba = beta + (long)a * Np;
    const double* __restrict__ bbv = beta + (long)bb * Np;
    const double* __restrict__ iK = invK + (long)a * Np * Np;
    __shared__ double red[4];
    // column-tile operands (shared by the 4 waves) are staged through LDS with a one-tile prefetch:
    // rows 0..7 Wt, row 8 Lb, row 9 beta_b  -> 640 doubles per tile
    __shared__ double Cs[2][EMK + 2][64];
    const int fr = lane & 15, fk = lane >> 4, i0 = ti * 64 + 16 * wave;
    // A fragments (constant over the sweep) and the row data of this lane's 4 accumulator rows
    const double a0 = o[(long)fk * Np + i0 + fr], a1 = o[(long)(4 + fk) * Np + i0 + fr];
    double la[4], bai[4];
    int irow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        irow[r] = i0 + crow(lane, r, crow_mode);
        la[r] = La[irow[r]];
        bai[r] = (irow[r] < N) ? ba[irow[r]] : 0.0;
    }
    double acc = 0.0;
    const int jt_end = diag ? ti + 1 : tiles;
    double st[3];
    auto fetch = [&](int jt) {   // 640 values / 256 threads: element e = tid + 256 q -> (row e / 64, col e % 64)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
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
        for (int q = 0; q < 3; ++q) {
            const int e = tid + 256 * q, rw = e >> 6, cl = e & 63;
            if (rw < EMK + 2) Cs[buf][rw][cl] = st[q];
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    int cur = 0;
    for (int jt = 0; jt < jt_end; ++jt) {
        if (jt + 1 < jt_end) fetch(jt + 1);
        const double mult = (diag && jt < ti) ? 2.0 : 1.0;
        double ik[4][4];
        if (diag) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ik[t][r] = iK[(long)irow[r] * Np + jt * 64 + 16 * t + fr];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cl = 16 * t + fr, j = jt * 64 + cl;
            d4 c = d4{0.0, 0.0, 0.0, 0.0};
            c = mfma16(a0, Cs[cur][fk][cl], c);
            c = mfma16(a1, Cs[cur][4 + fk][cl], c);
            const double lbj = Cs[cur][EMK][cl];
            const double bj = Cs[cur][EMK + 1][cl];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double q = exp_lean((la[r] + lbj) + c[r]);
                double wgt = bai[r] * bj;
                if (diag) wgt -= ik[t][r];
                acc += (j < N && irow[r] < N) ? mult * wgt * q : 0.0;
            }
        }
        if (jt + 1 < jt_end) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) partial[((long)b * P + p) * tiles + ti] = (red[0] + red[1]) + (red[2] + red[3]);
}

// cov_ab = t_p * sum_tiles partial;  cov_aa += sf_a^2;  cov -= mean mean^T;  symmetric fill.
// grid (ceil(B*P/64)), 64 threads: one thread per (input, pair).
__global__ void __launch_bounds__(64) em_finish_kernel(const double* __restrict__ partial, const double* __restrict__ prep,
                                                       const double* __restrict__ hyper, const double* __restrict__ mean,
                                                       double* __restrict__ cov, int B, int Ny, int d, int tiles) {
    const int P = Ny * (Ny + 1) / 2, stride = d * d + 1;
    const long gid = (long)blockIdx.x * 64 + threadIdx.x;
    if (gid >= (long)B * P) return;
    const int b = (int)(gid / P), p = (int)(gid % P);
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= p) ++a;
    const int bb = p - a * (a + 1) / 2;
    const double t = prep[((long)b * (Ny + P) + Ny + p) * stride + d * d];
    double s = 0.0;
    for (int k = 0; k < tiles; ++k) s += partial[((long)b * P + p) * tiles + k];
    double v = t * s;
    if (a == bb) v += hyper[(long)a * (d + 2) + d] * hyper[(long)a * (d + 2) + d];
    v -= mean[(long)b * Ny + a] * mean[(long)b * Ny + bb];
    cov[((long)b * Ny + a) * Ny + bb] = v;
    cov[((long)b * Ny + bb) * Ny + a] = v;
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
