// Tile-owner workers of the blocked Cholesky: the bulk of the factorisation (panel rows and trailing
// updates) as ONE persistent kernel next to the chain kernel (chol_chain.hpp), instead of two GEMM
// launches per panel step.
//
// The lower triangle is cut into 64 x 64 tiles (i, j), i >= j; the tiles with j >= 1 are numbered row by
// row, tile t belongs to worker t mod NW and LIVES IN THAT WORKER'S REGISTERS until its last update (at most
// WORKER_MAXT tiles per worker; 8 waves per worker, wave (wr, wc) holds the 16 x 32 piece (wr, wc) of every
// tile as two MFMA accumulators = 16 VGPRs per tile, 144 of the 256 a wave may use at two waves per SIMD).  The right-looking updates
// therefore cost no HBM traffic at all -- the trailing matrix is read once (from the K build) and never
// written back; what moves is the panel column L(:,k), 64 columns per step, through L2.
//
// Per panel step k every worker, in this order:
//   1. panel tiles (i,k), i >= k+2, it owns: wait leafdone[k]; L(i,k) = A(i,k) inv_kk^T; store to L;
//      count it in pancount[k] -- the last one raises colready[k] -- (and raise row2done[k] for i = k+2);
//   2. the two tiles the chain needs next, (k+2,k+1) and (k+2,k+2): as soon as L(k+2,k) [row2done] and
//      L(k+1,k) [pan1, the chain's own row] exist, update, store the tile to K and raise tdone[k][.];
//   3. all other live tiles (i,j), j > k: once the whole panel column is there (colready[k] and
//      pan1[k]): A(i,j) -= L(i,k) L(j,k)^T, operands staged through LDS; the operands of the next tile are
//      fetched into registers while the MFMAs of the current one run (8 waves = 2 per SIMD keep the fp64
//      MFMA pipe issuing every ~104 cycles; one wave per SIMD manages one per ~142; four would reach ~75
//      but leave only 128 registers per thread, less than the resident tiles need).
// Tiles (0,0), (1,0), (1,1) and, after their hand-off, (k+1,k) / (k+1,k+1) belong to the chain.
// Dead-lock freedom: a wait only ever targets work of an earlier step or a panel tile of the same step,
// and every worker does its panel tiles first; all NW workers plus the chain must be co-resident (one
// workgroup per CU: the LDS request below is > 80 KB), which the host guarantees by sizing NW to the CU
// count.  Every spin is bounded (wg_sync.hpp) and a time-out makes the host fall back to GEMM launches.
// A factorisation may be cut in two launches (kb, ksteps): the first covers the early steps with all tiles
// resident, the second the rest -- a quarter of the tiles -- with fewer workers, so that the CUs it leaves free
// can run the part of the triangular inverse that is already computable.
// grid (NW, 1, batch), 512 threads, dynamic LDS WORKER_LDS_BYTES.
#pragma once
#include "chol_chain.hpp"

namespace gpmpc {

constexpr int WORKER_MAXT = 9;
constexpr int WORKER_THREADS = 512;
constexpr int WORKER_EPT = 4096 / WORKER_THREADS;   // elements of a 64 x 64 block per thread
constexpr int WORKER_LDS_BYTES = 90000;   // two 64 x 65 operand blocks (66.6 KB); > 80 KB keeps one worker per CU

// acc[c] += sgn * A(16 rows at ar, 64 deep) * B(16 rows at br + 16 c, 64 deep)^T, c = 0, 1; operands in LDS
__device__ __forceinline__ void lds_mm_tile(const double* A, int ar, const double* B, int br, int lane, d4* acc, double sgn) {
    const int fr = lane & 15, fk = lane >> 4;
#pragma unroll 4
    for (int k0 = 0; k0 < 64; k0 += 4) {
        const double a = sgn * A[(ar + fr) * LS + k0 + fk];
        acc[0] = mfma16(a, B[(br + fr) * LS + k0 + fk], acc[0]);
        acc[1] = mfma16(a, B[(br + 16 + fr) * LS + k0 + fk], acc[1]);
    }
}

// a 64 x 64 block between global memory, WORKER_EPT registers per thread, and LDS.  Addresses are a
// workgroup-uniform base plus ONE per-thread unsigned 32-bit BYTE offset (8 (row * ld + col) of the thread's
// first element) so that the accesses can take the SGPR-base form and the addresses cost one register.
__device__ __forceinline__ const double& at_byte(const double* base, unsigned byte_off) {
    return *(const double*)((const char*)base + byte_off);
}
__device__ __forceinline__ double& at_byte(double* base, unsigned byte_off) { return *(double*)((char*)base + byte_off); }

// (the thread's i-th element sits 8 i rows below its first: a uniform pointer step, one offset register)
__device__ __forceinline__ void block_to_regs(double* r, const double* __restrict__ src, unsigned toff0, long ld) {
#pragma unroll
    for (int i = 0; i < WORKER_EPT; ++i) r[i] = at_byte(src + (long)(WORKER_THREADS / 64) * i * ld, toff0);
}
__device__ __forceinline__ void regs_to_lds(double* dst, const double* r, int tid) {
#pragma unroll
    for (int i = 0; i < WORKER_EPT; ++i) {
        const int idx = tid + WORKER_THREADS * i, rr = idx >> 6, cc = idx & 63;
        dst[rr * LS + cc] = r[i];
    }
}

__global__ void __launch_bounds__(WORKER_THREADS) chol_worker_kernel(double* Kmat, double* L, const double* Inv, long ld,
                                                                     long sBatch, int nb_all, int* flags, long sFlags,
                                                                     int crow_mode, int spin_limit, int kb, int ksteps,
                                                                     int* ready) {
    double* smem = GPMPC_DYN_SMEM();
    double* A = smem;
    double* B = A + 64 * LS;
    int* slot = (int*)(B + 64 * LS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;                 // this wave's 16 x 32 piece: rows 16 wr, columns 32 wc
    const int w = blockIdx.x, NW = gridDim.x;
    // A launch works on the trailing matrix from block kb on, for `ksteps` panel steps: everything below is
    // written for kb = 0 and made relative by shifting the base pointers and the flag arrays by kb.
    const int nb = nb_all - kb;
    const long mb = (long)blockIdx.z * sBatch + (long)(64 * kb) * ld + 64 * kb;
    double* __restrict__ Kb = Kmat + mb;
    double* __restrict__ Lb = L + mb;
    const double* __restrict__ Ib = Inv + mb;
    int* fl = flags + (long)blockIdx.z * sFlags;
    int* err = fl;
    int* leafdone = fl + 1 + kb;
    int* pan1 = fl + 1 + nb_all + kb;
    int* tdone = fl + 1 + 2 * nb_all + 2 * kb;
    int* pancount = fl + 1 + 4 * nb_all + kb;
    int* row2done = fl + 1 + 5 * nb_all + kb;
    int* colready = fl + 1 + 6 * nb_all + kb;
    int* progress = fl + 1 + 7 * nb_all + (w & 255);
    if (tid == 0) flag_store(progress, 1);
    if (ready && tid == 0) {                       // "all workgroups of this launch are resident" for the host's gates
        int* rd = ready + (long)blockIdx.z * sFlags;
        if (__hip_atomic_fetch_add(rd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == NW) flag_store(rd + 1, 1);
    }
#ifndef GPMPC_EMULATED
    if (tid == 0) flag_store(progress + 256, (int)((wall_clock64() / 100) & 0x3fffffff));   // start time, us
#endif
    const int ntiles = (nb - 1) * nb / 2 - 1;    // resident tiles: (i, j), 1 <= j <= i, without (1,1)
    // per-thread BYTE offsets inside a 64 x 64 tile of an [ld]-strided matrix: csub = the thread's first
    // accumulator element (sub-tile row crow(lane, 0), column lane & 15), toff = its first copy element
    // (crow(lane, r) is linear in r with a uniform step -- 4 rows in the gfx950 f64 map -- so element r sits
    //  r * cstep rows below element 0: one offset register, the step goes into the uniform base pointer)
    const long cstep = (long)(crow(0, 1, crow_mode) - crow(0, 0, crow_mode)) * ld;
    const unsigned csub = 8u * (unsigned)((16 * wr + crow(lane, 0, crow_mode)) * (int)ld + 32 * wc + (lane & 15));
    const unsigned toff = 8u * (unsigned)((tid >> 6) * (int)ld + (tid & 63));

    // my tiles: (ti[n], tj[n]) in LDS (workgroup-uniform; ti < 0 = none / finished), the tiles themselves in
    // registers C[n].  The slot index is a run-time value: the code that touches a tile is
    // addressed through a switch over compile-time indices so that C[] never becomes an indexed (scratch) array.
    int* ti = slot + 4;
    int* tj = ti + WORKER_MAXT;
    d4 C[WORKER_MAXT][2];
    if (tid < WORKER_MAXT) {
        const int t = w + tid * NW;
        int i = -1, j = 0;
        if (t < ntiles) {                             // t + 1 enumerates the triangle (r, c), c <= r, of (i-1, j-1)
            int r = 0;
            while ((r + 1) * (r + 2) / 2 <= t + 1) ++r;
            i = r + 1;
            j = t + 1 - r * (r + 1) / 2 + 1;
        }
        ti[tid] = i; tj[tid] = j;
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < WORKER_MAXT; ++n) {
        const int i = __builtin_amdgcn_readfirstlane(ti[n]), j = __builtin_amdgcn_readfirstlane(tj[n]);
        C[n][0] = C[n][1] = d4{0.0, 0.0, 0.0, 0.0};
        if (i >= 0) {
            const double* src = Kb + (long)(64 * i) * ld + 64 * j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                C[n][0][r] = at_byte(src + r * cstep, csub);
                C[n][1][r] = at_byte(src + r * cstep, csub + 128u);
            }
        }
    }

    int k = 0;
    for (; k + 2 < nb && k < ksteps; ++k) {
        // ---- 1. panel tiles of column k.  Column 0 never receives an update, so its tiles are not kept in
        //         registers: at k = 0 worker w takes rows 2 + w, 2 + w + NW, ... straight from K.
        for (int i = 2 + w; k == 0 && i < nb; i += NW) {
            if (tid == 0) flag_store(progress, 1 + 4 * k + 1);
            if (!wg_wait2(&leafdone[0], 1, nullptr, 0, err, spin_limit, slot, 2000000 + w)) return;
            double ra[WORKER_EPT], rb[WORKER_EPT];
            block_to_regs(ra, Kb + (long)(64 * i) * ld, toff, ld);
            block_to_regs(rb, Ib, toff, ld);                                        // inv_00
            regs_to_lds(A, ra, tid);
            regs_to_lds(B, rb, tid);
            __syncthreads();
            d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
            lds_mm_tile(A, 16 * wr, B, 32 * wc, lane, acc, 1.0);
            double* dst = Lb + (long)(64 * i) * ld;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                at_byte(dst + r * cstep, csub) = acc[0][r];
                at_byte(dst + r * cstep, csub + 128u) = acc[1][r];
            }
            GPMPC_DRAIN_VM();
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                GPMPC_DRAIN_VM();
                if (i == 2) flag_store(&row2done[0], 1);
                const int before = __hip_atomic_fetch_add(&pancount[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (before + 1 == nb - 2) flag_store(&colready[0], 1);
            }
            __syncthreads();
        }
#pragma unroll
        for (int n = 0; n < WORKER_MAXT; ++n) {
            const int i = __builtin_amdgcn_readfirstlane(ti[n]), j = __builtin_amdgcn_readfirstlane(tj[n]);
            if (i < 0 || j != k) continue;             // (workgroup-uniform)
            if (tid == 0) flag_store(progress, 1 + 4 * k + 1);
            if (!wg_wait2(&leafdone[k], 1, nullptr, 0, err, spin_limit, slot, 2000000 + 1000 * k + w)) return;
            double rb[WORKER_EPT];
            block_to_regs(rb, Ib + (long)(64 * k) * ld + 64 * k, toff, ld);        // inv_kk (zeros above the diagonal)
            lds_put16(A, 16 * wr, 32 * wc, C[n][0], 1.0, lane, crow_mode);
            lds_put16(A, 16 * wr, 32 * wc + 16, C[n][1], 1.0, lane, crow_mode);
            regs_to_lds(B, rb, tid);
            __syncthreads();
            d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
            lds_mm_tile(A, 16 * wr, B, 32 * wc, lane, acc, 1.0);
            double* dst = Lb + (long)(64 * i) * ld + 64 * k;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                at_byte(dst + r * cstep, csub) = acc[0][r];
                at_byte(dst + r * cstep, csub + 128u) = acc[1][r];
            }
            GPMPC_DRAIN_VM();
            __syncthreads();                           // (also: everybody is done with A and B)
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                GPMPC_DRAIN_VM();
                if (i == k + 2) flag_store(&row2done[k], 1);
                // count the tile; whoever completes the column raises the flag the consumers poll
                const int before = __hip_atomic_fetch_add(&pancount[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (before + 1 == nb - k - 2) flag_store(&colready[k], 1);
                ti[n] = -1;
            }
            __syncthreads();
        }
        // ---- 2. (k+2,k+1), (k+2,k+2) for the chain
#pragma unroll
        for (int n = 0; n < WORKER_MAXT; ++n) {
            const int i = __builtin_amdgcn_readfirstlane(ti[n]), j = __builtin_amdgcn_readfirstlane(tj[n]);
            if (i != k + 2 || j <= k) continue;         // j is k+1 or k+2
            if (tid == 0) flag_store(progress, 1 + 4 * k + 2);
            if (!wg_wait2(&row2done[k], 1, j == k + 1 ? &pan1[k] : nullptr, 1, err, spin_limit, slot, 3000000 + 1000 * k + w))
                return;
            double ra[WORKER_EPT], rb[WORKER_EPT];
            block_to_regs(ra, Lb + (long)(64 * (k + 2)) * ld + 64 * k, toff, ld);
            block_to_regs(rb, Lb + (long)(64 * j) * ld + 64 * k, toff, ld);
            regs_to_lds(A, ra, tid);
            regs_to_lds(B, rb, tid);
            __syncthreads();
            double* dst = Kb + (long)(64 * (k + 2)) * ld + 64 * j;
            lds_mm_tile(A, 16 * wr, B, 32 * wc, lane, C[n], -1.0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                at_byte(dst + r * cstep, csub) = C[n][0][r];
                at_byte(dst + r * cstep, csub + 128u) = C[n][1][r];
            }
            wg_publish(&tdone[2 * k + (j == k + 2 ? 1 : 0)], 1);
            if (tid == 0) ti[n] = -1;
            __syncthreads();                           // A, B free again; ti[] visible
        }
        // ---- 3. all other live tiles, software-pipelined: operands of the next tile travel global -> registers
        //         while the MFMAs of the current one run from LDS
        int cur = -1;
        for (int n = WORKER_MAXT - 1; n >= 0; --n)
            if (__builtin_amdgcn_readfirstlane(ti[n]) >= 0 && __builtin_amdgcn_readfirstlane(tj[n]) > k) cur = n;
        if (cur < 0) continue;
        if (tid == 0) flag_store(progress, 1 + 4 * k + 3);
        if (!wg_wait2(&colready[k], 1, &pan1[k], 1, err, spin_limit, slot, 4000000 + 1000 * k + w)) return;
        double ra[WORKER_EPT], rb[WORKER_EPT];
        block_to_regs(ra, Lb + (long)(64 * __builtin_amdgcn_readfirstlane(ti[cur])) * ld + 64 * k, toff, ld);
        block_to_regs(rb, Lb + (long)(64 * __builtin_amdgcn_readfirstlane(tj[cur])) * ld + 64 * k, toff, ld);
#pragma unroll
        for (int n = 0; n < WORKER_MAXT; ++n) {
            if (n != cur) continue;                    // (slots before the first / between live tiles)
            regs_to_lds(A, ra, tid);
            regs_to_lds(B, rb, tid);
            __syncthreads();
            int nxt = -1;
            for (int m = WORKER_MAXT - 1; m > n; --m)
                if (__builtin_amdgcn_readfirstlane(ti[m]) >= 0 && __builtin_amdgcn_readfirstlane(tj[m]) > k) nxt = m;
            if (nxt >= 0) {
                block_to_regs(ra, Lb + (long)(64 * __builtin_amdgcn_readfirstlane(ti[nxt])) * ld + 64 * k, toff, ld);
                block_to_regs(rb, Lb + (long)(64 * __builtin_amdgcn_readfirstlane(tj[nxt])) * ld + 64 * k, toff, ld);
            }
            lds_mm_tile(A, 16 * wr, B, 32 * wc, lane, C[n], -1.0);
            __syncthreads();                           // A, B free again
            cur = nxt;
        }
    }
    // a launch that stops before the last step hands its live tiles back through K: the next launch (from block
    // kb + ksteps on, with fewer workers -- the freed CUs take the inverse pipeline) reloads them
    if (k + 2 < nb) {
#pragma unroll
        for (int n = 0; n < WORKER_MAXT; ++n) {
            const int i = __builtin_amdgcn_readfirstlane(ti[n]), j = __builtin_amdgcn_readfirstlane(tj[n]);
            if (i < 0) continue;
            double* dst = Kb + (long)(64 * i) * ld + 64 * j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                at_byte(dst + r * cstep, csub) = C[n][0][r];
                at_byte(dst + r * cstep, csub + 128u) = C[n][1][r];
            }
        }
    }
}

}  // namespace gpmpc
