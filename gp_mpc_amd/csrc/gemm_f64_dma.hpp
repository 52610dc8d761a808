// fp64 GEMM (128 x 128 and 64 x 64 tiles) with operand tiles streamed HBM/L2 -> LDS by the DMA path of the load
// unit (buffer_load_dwordx4 ... lds, lds_dma.hpp): no staging registers, no LDS store instructions and no VALU work
// in the K loop.
//
// Same contract as gemm_f64_kernel (gemm_f64.hpp, GemmP):  C = alpha * A B + beta * C  with either operand stored
// K-contiguous or M/N-contiguous, the triangular K ranges, the `lower` output mask, the chain hand-off flags and the
// fused column-sum-of-squares epilogue of the predictive variance (a9, gp_functions.py:122-126).  Preconditions
// (gemm_dma_supported; the launcher keeps the register-staged kernel otherwise): K a multiple of 16, 16-byte
// aligned operands with even leading dimensions, each operand below 4 GB.
//
// K-contiguous operand, A(m,k) = A[m*lda + k]:
// LDS image of one K slab (16 doubles = 128 bytes per row): [128 rows of A][128 rows of B], row r at byte 128 r,
// and inside a row the eight 16-byte pieces are permuted, piece c stored at position c ^ ((r >> 1) & 7).  A DMA
// load writes wave-uniform base + 16 lane, so one wave instruction fills 8 consecutive rows (1 KB) and the
// permutation is applied on the SOURCE side (each lane fetches the piece that belongs at its position).  An MFMA
// fragment read is one ds_read_b128 per 16-row tile: lane (row = l & 15, q = l >> 4) takes piece q (or 4 + q) of its
// row -- two consecutive K values used by two successive matrix instructions -- and the 16 lanes that are served
// together hit 16 different bank groups (rows r, r+1 differ in bit 5 of the bank index, pairs of rows in the
// permuted piece position).  Which K index a lane group holds in a given instruction does not matter for the
// product as long as A and B fragments agree, which they do by construction.
// M/N-contiguous operand, A(m,k) = A[k*lda + m]: the image is [16 K rows][tile width], every other pair of K rows with
// its two 16-element halves swapped (element m at m ^ 16: the four lane groups of a ds_read_b64 fragment read then
// alternate between the two 128-byte bank halves); one wave instruction fills 1 KB of consecutive K rows.
// A lower-triangular A: inside the diagonal block, a wave whose rows all lie above a slab's K range skips that slab.
//
// Pipeline: STAGES slab images form a ring; slab t + STAGES - 1 is requested right after the barrier of step t
// (its image was last read in step t - 1), the wait in front of the barrier of step t is a COUNTED s_waitcnt
// vmcnt that leaves the younger slabs in flight.  The loads are inline assembly: with the compiler's own
// DMA builtin every LDS read is preceded by vmcnt(0), which drains the ring (measured in the assembly output).
#pragma once
#include "gemm_f64.hpp"
#include "lds_dma.hpp"

namespace gpmpc {


// WPS: waves per SIMD the register allocation must allow (workgroups per CU x waves per workgroup / 4)
// AMC / BNC: operand stored with M (resp. N) contiguous instead of K, as in GemmP::a_mc / b_nc
template <int BM, int BN, int WGM, int WGN, int STAGES, int WPS, bool AMC, bool BNC>
__global__ void __launch_bounds__(64 * WGM * WGN, WPS) gemm_f64_dma_kernel(GemmP p) {
    constexpr int BK = 16;
    constexpr int NW = WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
    constexpr int LA = BM / 8, LB = BN / 8;            // wave loads (1 KB each) per slab image
    constexpr int LPW = (LA + LB) / NW;                // ... per wave
    constexpr int IMG_A = BM * 128, SLAB = (BM + BN) * 128;
    constexpr int RBA = BM * 8, RBB = BN * 8;          // M/N-contiguous images: bytes per K row
    static_assert((LA + LB) % NW == 0 && STAGES >= 2 && STAGES <= 5, "bad configuration");
    char* smem = (char*)GPMPC_DYN_SMEM();
    int* slot = reinterpret_cast<int*>(smem + STAGES * SLAB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
    const int tme = (int)blockIdx.x / p.npad, tne = (int)blockIdx.x % p.npad;
    if (tme >= p.tilesMe || tne >= p.tilesNe) return;
    if (p.wait_flag) {
        if (!wg_wait2(p.wait_flag + (long)blockIdx.z * p.sFlags, 1, nullptr, 0, p.err + (long)blockIdx.z * p.sFlags,
                      p.spin_limit, slot))
            return;
    }

    const int z1 = p.zdiv > 0 ? (int)blockIdx.z % p.zdiv : (p.zmap ? p.zmap[blockIdx.z] : (int)blockIdx.z);
    const int z2 = p.zdiv > 0 ? (p.zmap ? p.zmap[(int)blockIdx.z / p.zdiv] : (int)blockIdx.z / p.zdiv) : 0;
    const double* __restrict__ A = p.A + (long)z1 * p.sA + (long)z2 * p.sA2;
    const double* __restrict__ B = p.B + (long)z1 * p.sB + (long)z2 * p.sB2;
    const dma_rsrc_t rsA = dma_make_rsrc(A, (unsigned)((long)(AMC ? p.K : p.M) * p.lda * 8));
    const dma_rsrc_t rsB = dma_make_rsrc(B, (unsigned)((long)(BNC ? p.K : p.N) * p.ldb * 8));
    const int fr = lane & 15, fq = lane >> 4;
    // Fragment read offsets inside a slab image.  The four matrix instructions of a slab half h use, in lane group q,
    // K index 8h + 2q + e (e = 0, 1) of BOTH operands.
    //   K-contiguous image: 16-byte piece c of row r at r*128 + ((c ^ ((r>>1)&7)) << 4); one ds_read_b128 fetches
    //     (e = 0, e = 1); tile i is 2048 bytes further; h flips bit 6 of the offset.
    //   M/N-contiguous image: K row k at k*RB, element m at ((m ^ (16 * ((k>>1)&1))) * 8: K rows 2q of the four
    //     lane groups alternate between the two 128-byte bank halves; one ds_read_b64 per (h, e) at +(8h+e)*RB.
    const unsigned sw = (unsigned)((fr >> 1) & 7);
    unsigned fa[AMC ? TM : 2], fb[BNC ? TN : 2];
    if (AMC) {
#pragma unroll
        for (int i = 0; i < (AMC ? TM : 0); ++i)
            fa[i] = (unsigned)(2 * fq * RBA + (((wm * WM + 16 * i + fr) ^ (16 * (fq & 1))) << 3));
    } else {
        fa[0] = (unsigned)((wm * WM + fr) * 128) + (((unsigned)fq ^ sw) << 4);
        fa[1] = fa[0] ^ 64u;
    }
    if (BNC) {
#pragma unroll
        for (int j = 0; j < (BNC ? TN : 0); ++j)
            fb[j] = (unsigned)(IMG_A + 2 * fq * RBB + (((wn * WN + 16 * j + fr) ^ (16 * (fq & 1))) << 3));
    } else {
        fb[0] = (unsigned)(IMG_A + (wn * WN + fr) * 128) + (((unsigned)fq ^ sw) << 4);
        fb[1] = fb[0] ^ 64u;
    }

    for (int pass = 0; pass < 2; ++pass) {
        int tm = (p.kflags & KA_LE_M) ? tilesM - 1 - tme : tme;     // heavy tiles first
        int tn = (p.kflags & KB_LE_N) ? tilesN - 1 - tne : tne;
        if (pass == 1) {
            if (p.pair == PAIR_M) {
                if (tilesM - 1 - tm == tm) break;
                tm = tilesM - 1 - tm;
            } else if (p.pair == PAIR_N) {
                if (tilesN - 1 - tn == tn) break;
                tn = tilesN - 1 - tn;
            } else {
                break;
            }
        }
        const int m0 = tm * BM, n0 = tn * BN;
        if (p.lower && n0 > m0 + BM - 1) continue;
        if (p.skip00 && tm == 0 && tn == 0) continue;

        int klo = 0, khi = p.K;
        if (p.kflags & KA_LE_M) khi = min(khi, m0 + BM);
        if (p.kflags & KA_GE_M) klo = max(klo, m0);
        if (p.kflags & KB_LE_N) khi = min(khi, n0 + BN);
        if (p.kflags & KB_GE_N) klo = max(klo, n0);
        klo = klo / BK * BK;
        khi = min(p.K, (khi + BK - 1) / BK * BK);
        const int nk = khi > klo ? (khi - klo) / BK : 0;

        // This lane's source offsets for its LPW pieces of a slab (the slab's K offset rides in the scalar offset).
        // Rows (K-contiguous) or columns (M/N-contiguous) beyond the operand are clamped: what they fetch only
        // reaches rows / columns of the product that are never stored (the column sums mask them).
        unsigned vo[LPW];
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int j = wave + NW * i;                           // [0, LA): piece j of the A image, else of the B image
            const bool isA = j < LA;
            const int jj = isA ? j : j - LA;
            if (isA ? AMC : BNC) {
                const int rb = isA ? RBA : RBB;
                const int byte = 1024 * jj + 16 * lane, k = byte / rb, pos = (byte % rb) >> 4;
                const int piece = pos ^ (8 * ((k >> 1) & 1));
                const long ldx = isA ? p.lda : p.ldb;
                const long col = min((long)(isA ? m0 : n0) + 2 * piece, ldx - 2);     // stay inside the K row
                vo[i] = (unsigned)(((long)k * ldx + col) * 8);
            } else {
                const int row = 8 * jj + (lane >> 3);
                const unsigned piece = (unsigned)(lane & 7) ^ (unsigned)((row >> 1) & 7);
                const long grow = isA ? (long)min(m0 + row, p.M - 1) * p.lda : (long)min(n0 + row, p.N - 1) * p.ldb;
                vo[i] = (unsigned)(grow * 8) + (piece << 4);
            }
        }
        const unsigned kstepA = AMC ? (unsigned)(p.lda * 8) : 8u, kstepB = BNC ? (unsigned)(p.ldb * 8) : 8u;
        const lds_addr_t smem_at = lds_addr_of(smem);
        auto request = [&](int t) {                               // slab t of this tile -> ring image t % STAGES
            const lds_addr_t img = smem_at + (lds_addr_t)((t % STAGES) * SLAB);
            const unsigned k0 = (unsigned)(klo + t * BK);
#pragma unroll
            for (int i = 0; i < LPW; ++i) {
                const int j = wave + NW * i;
                if (j < LA) dma_load16(rsA, img + 1024 * j, vo[i], k0 * kstepA);
                else dma_load16(rsB, img + 1024 * j, vo[i], k0 * kstepB);
            }
        };

        d4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

        auto multiply = [&](int t) {
            const char* img = smem + (t % STAGES) * SLAB;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double2 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (AMC) {
                        a[i].x = *reinterpret_cast<const double*>(img + fa[AMC ? i : 0] + (8 * h) * RBA);
                        a[i].y = *reinterpret_cast<const double*>(img + fa[AMC ? i : 0] + (8 * h + 1) * RBA);
                    } else {
                        a[i] = *reinterpret_cast<const double2*>(img + fa[AMC ? 0 : h] + i * 2048);
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (BNC) {
                        b[j].x = *reinterpret_cast<const double*>(img + fb[BNC ? j : 0] + (8 * h) * RBB);
                        b[j].y = *reinterpret_cast<const double*>(img + fb[BNC ? j : 0] + (8 * h + 1) * RBB);
                    } else {
                        b[j] = *reinterpret_cast<const double2*>(img + fb[BNC ? 0 : h] + j * 2048);
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(a[i].x, b[j].x, acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(a[i].y, b[j].y, acc[i][j]);
            }
        };

        // Lower-triangular A: inside the diagonal block the slabs whose K offset lies beyond this wave's last row
        // multiply exact zeros; the wave sits them out (it still requests its pieces and meets the barriers) and
        // the matrix pipe time goes to the waves that have work.
        const int kskip = (p.kflags & KA_LE_M) ? (m0 + (wm + 1) * WM - klo + BK - 1) / BK : (1 << 30);
        // prologue: STAGES - 1 slabs requested (requests past the end are skipped: the tail waits for everything)
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            if (t < nk) request(t);
        int kt = 0;
        for (; kt + STAGES - 1 < nk; ++kt) {                      // steady state: slab kt + STAGES - 1 exists
            dma_wait<LPW * (STAGES - 2)>();                       // slab kt has landed (this wave's pieces)
            dma_barrier();                                        // ... everybody's, and image (kt - 1) is free
            request(kt + STAGES - 1);
            if (kt < kskip) multiply(kt);
        }
        for (; kt < nk; ++kt) {
            dma_wait<0>();
            dma_barrier();
            if (kt < kskip) multiply(kt);
        }
        dma_barrier();                                            // all fragment reads done: LDS reusable

        if (p.epi == EPI_STORE) {
            double* __restrict__ C = p.C + (long)z1 * p.sC + (long)z2 * p.sC2;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + wm * WM + i * 16 + crow(lane, r, p.crow_mode);
                        const int n = n0 + wn * WN + j * 16 + fr;
                        if (m < p.M && n < p.N && (!p.lower || n <= m)) {
                            double* c = C + (long)m * p.ldc + n;
                            double v = p.alpha * acc[i][j][r];
                            if (p.beta != 0.0) v += p.beta * (*c);
                            *c = v;
                        }
                    }
        } else {
            // column sums of squares over this tile's rows < M
            double* red = reinterpret_cast<double*>(smem);         // [WGM][BN]
            if (p.Ct) gemm_store_transposed<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + wm * WM + i * 16 + crow(lane, r, p.crow_mode);
                        const double v = (m < p.M) ? acc[i][j][r] : 0.0;
                        s += v * v;
                    }
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                if (lane < 16) red[wm * BN + wn * WN + j * 16 + lane] = s;
            }
            __syncthreads();
            if (tid < BN && n0 + tid < p.N) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < WGM; ++w) t += red[w * BN + tid];
                p.part[(long)blockIdx.z * p.sPart + (long)tm * p.ldpart + n0 + tid] = t;
            }
        }
        if (p.done_flags && tm == 1 && tn <= 1) wg_publish(p.done_flags + (long)blockIdx.z * p.sFlags + tn, 1);
        __syncthreads();  // LDS (red) is reused by the next pass
    }
}

// true if gemm_f64_dma_kernel can run this product
inline bool gemm_dma_supported(const GemmP& p) {
    const long rowsA = p.a_mc ? p.K : p.M, rowsB = p.b_nc ? p.K : p.N;
    return !p.remap && p.K % 16 == 0 && p.lda % 2 == 0 && p.ldb % 2 == 0 && ((unsigned long)p.A & 15) == 0 &&
           ((unsigned long)p.B & 15) == 0 && (p.sA % 2 == 0) && (p.sB % 2 == 0) && (p.sA2 % 2 == 0) && (p.sB2 % 2 == 0) &&
           rowsA * p.lda * 8 < (1L << 32) && rowsB * p.ldb * 8 < (1L << 32);
}

template <int BM, int BN, int WGM, int WGN, int STAGES, int WPS, bool AMC, bool BNC>
inline void launch_gemm_dma_kernel(const GemmP& p, dim3 grid, hipStream_t stream) {
    constexpr int lds = STAGES * (BM + BN) * 128 + 16;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_dma_kernel<BM, BN, WGM, WGN, STAGES, WPS, AMC, BNC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f64_dma_kernel<BM, BN, WGM, WGN, STAGES, WPS, AMC, BNC>), grid, dim3(64 * WGM * WGN), lds,
                       stream, p);
}

template <int BM, int BN, int WGM, int WGN, int STAGES, int WPS>
inline void launch_gemm_dma(GemmP p, int batch, hipStream_t stream, int resident, int min_pair_blocks = 512) {
    const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
    p.pair = PAIR_NONE;
    p.tilesMe = tilesM;
    p.tilesNe = tilesN;
    const bool one_flag = (p.kflags == KA_LE_M || p.kflags == KA_GE_M || p.kflags == KB_LE_N || p.kflags == KB_GE_N);
    const long nblocks = (long)tilesM * tilesN * batch;
    if (one_flag && !p.lower && nblocks <= (long)resident && nblocks >= min_pair_blocks) {
        if ((p.kflags & (KA_LE_M | KA_GE_M)) && tilesM >= 2) {
            p.pair = PAIR_M;
            p.tilesMe = (tilesM + 1) / 2;
        } else if ((p.kflags & (KB_LE_N | KB_GE_N)) && tilesN >= 2) {
            p.pair = PAIR_N;
            p.tilesNe = (tilesN + 1) / 2;
        }
    }
    // XCD-consistent column residues: from 16 tile columns on, the grid width is padded to a multiple of 8 (consecutive
    // workgroups go round-robin to the 8 XCDs, so every XCD then keeps seeing the same B panels in its own L2).  With
    // the register-staged kernel this cost 1-3 % (gemm_f64.hpp); with the DMA-staged one it gains 0.5 % on the
    // variance GEMM and on the step.  GPMPC_PAD_MIN=<tiles> moves the threshold.
    static const int pad_min = getenv("GPMPC_PAD_MIN") ? atoi(getenv("GPMPC_PAD_MIN")) : 16;
    p.npad = (p.tilesNe >= pad_min) ? ((p.tilesNe + 7) & ~7) : p.tilesNe;
    const dim3 grid(p.tilesMe * p.npad, 1, batch);
    if (!p.a_mc && !p.b_nc) launch_gemm_dma_kernel<BM, BN, WGM, WGN, STAGES, WPS, false, false>(p, grid, stream);
    else if (!p.a_mc && p.b_nc) launch_gemm_dma_kernel<BM, BN, WGM, WGN, STAGES, WPS, false, true>(p, grid, stream);
    else if (p.a_mc && !p.b_nc) launch_gemm_dma_kernel<BM, BN, WGM, WGN, STAGES, WPS, true, false>(p, grid, stream);
    else launch_gemm_dma_kernel<BM, BN, WGM, WGN, STAGES, WPS, true, true>(p, grid, stream);
}

// Returns the tile edge used (the caller of EPI_COLSUMSQ sizes `part` with it).
inline int g_gemm_force_tile = 0;   // gpmpc_set_tuning("gemm_tile", ...): tests reach the large tiles with small matrices

inline int launch_gemm(const GemmP& p, int batch, hipStream_t stream, int force_tile = 0) {
    const int tile = force_tile ? force_tile : g_gemm_force_tile ? g_gemm_force_tile : gemm_pick_tile(p, batch);
    // GPMPC_GEMM_DMA: bit 0 the 128 x 128 tile, bit 1 the 64 x 64 tile, bit 2 the 32 x 32 tile (four-image ring: the
    // latency-bound small products of the inverse tree, -18 us on the C2 fit) through the DMA-staged kernel; default all
    static const int use_dma = getenv("GPMPC_GEMM_DMA") ? atoi(getenv("GPMPC_GEMM_DMA")) : 7;
    if (tile == 128 && (use_dma & 1) && gemm_dma_supported(p)) {
        launch_gemm_dma<128, 128, 2, 4, 2, 4>(p, batch, stream, 512);
    } else if (tile == 128) {
        launch_gemm_cfg<128, 128, 16, 2, 4>(p, batch, stream, 512);
    } else if (tile == 64 && (use_dma & 2) && gemm_dma_supported(p)) {
        // long K (the products of the inverse tree): three slab images; short K (the K = 64 updates of the flagged
        // factorisation, bound by their C traffic): two, so that more workgroups fit a CU  (C2 fit -15 us with three,
        // C3 fit +8 ms with three everywhere)
        static const int st64 = getenv("GPMPC_T64_STAGES") ? atoi(getenv("GPMPC_T64_STAGES")) : 0;
        if (st64 == 3 || (st64 == 0 && p.K >= 256)) launch_gemm_dma<64, 64, 2, 2, 3, 4>(p, batch, stream, 1024);
        else launch_gemm_dma<64, 64, 2, 2, 2, 4>(p, batch, stream, 1024);
    } else if (tile == 64) {
        launch_gemm_cfg<64, 64, 16, 2, 2>(p, batch, stream, 1024);
    } else if ((use_dma & 4) && gemm_dma_supported(p)) {
        launch_gemm_dma<32, 32, 2, 1, 4, 4>(p, batch, stream, 1024);
    } else if (p.K % 32 == 0) {
        launch_gemm_cfg<32, 32, 32, 2, 2>(p, batch, stream, 1024);
    } else {
        launch_gemm_cfg<32, 32, 16, 2, 2>(p, batch, stream, 1024);
    }
    return tile;
}

}  // namespace gpmpc
