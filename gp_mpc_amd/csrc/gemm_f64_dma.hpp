// fp64 GEMM, 128 x 128 tile, with operand tiles streamed HBM/L2 -> LDS by the DMA path of the load unit
// (buffer_load_dwordx4 ... lds): no staging registers, no LDS store instructions and no VALU work in the K loop.
//
// Same contract as gemm_f64_kernel (gemm_f64.hpp, GemmP) for the K-contiguous / K-contiguous case
//     C = alpha * A B + beta * C,   A(m,k) = A[m*lda + k],  B(k,n) = B[n*ldb + k]
// including the triangular K ranges and the fused column-sum-of-squares epilogue of the predictive variance
// (a9, gp_functions.py:122-126).  Not supported here (the launcher keeps the register-staged kernel for them):
// transposed operands, the chain hand-off flags, K not a multiple of 16, operands of 4 GB or more.
//
// LDS image of one K slab (16 doubles = 128 bytes per row): [128 rows of A][128 rows of B], row r at byte 128 r,
// and inside a row the eight 16-byte pieces are permuted, piece c stored at position c ^ ((r >> 1) & 7).  A DMA
// load writes wave-uniform base + 16 lane, so one wave instruction fills 8 consecutive rows (1 KB) and the
// permutation is applied on the SOURCE side (each lane fetches the piece that belongs at its position).  An MFMA
// fragment read is one ds_read_b128 per 16-row tile: lane (row = l & 15, q = l >> 4) takes piece q (or 4 + q) of its
// row -- two consecutive K values used by two successive matrix instructions -- and the 16 lanes that are served
// together hit 16 different bank groups (rows r, r+1 differ in bit 5 of the bank index, pairs of rows in the
// permuted piece position).  Which K index a lane group holds in a given instruction does not matter for the
// product as long as A and B fragments agree, which they do by construction.
//
// Pipeline: STAGES slab images form a ring; slab t + STAGES - 1 is requested right after the barrier of step t
// (its image was last read in step t - 1), the wait in front of the barrier of step t is a COUNTED s_waitcnt
// vmcnt that leaves the younger slabs in flight.  The loads are inline assembly: with the compiler's own
// DMA builtin every LDS read is preceded by vmcnt(0), which drains the ring (measured in the assembly output).
#pragma once
#include "gemm_f64.hpp"

namespace gpmpc {

constexpr int DMA_SLAB_BYTES = 2 * 128 * 128;   // one K slab of both operands

#ifdef GPMPC_EMULATED
struct dma_rsrc_t { const char* base; unsigned bytes; };
inline dma_rsrc_t dma_make_rsrc(const void* base, unsigned bytes) { return dma_rsrc_t{(const char*)base, bytes}; }
// wave-uniform LDS destination + 16 * lane
inline void dma_load16(const dma_rsrc_t& r, char* lds_wave_base, unsigned voff, unsigned soff) {
    const unsigned long o = (unsigned long)voff + soff;
    char* dst = lds_wave_base + 16 * (threadIdx.x & 63);
    if (o + 16 > r.bytes) { for (int i = 0; i < 16; ++i) dst[i] = 0; return; }
    for (int i = 0; i < 16; ++i) dst[i] = r.base[o + i];
}
template <int N> inline void dma_wait() {}
inline void dma_barrier() { __syncthreads(); }
#else
typedef int dma_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma_rsrc_t dma_make_rsrc(const void* base, unsigned bytes) {
    const unsigned long b = (unsigned long)base;
    dma_rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));   // stride 0, no swizzle
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void dma_load16(dma_rsrc_t r, char* lds_wave_base, unsigned voff, unsigned soff) {
    const unsigned dst = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)lds_wave_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(__builtin_amdgcn_readfirstlane(dst)), "v"(voff), "s"(r), "s"(soff)
                 : "memory");
}
// at most N of this wave's DMA loads still in flight
template <int N> __device__ __forceinline__ void dma_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0f70);
}
__device__ __forceinline__ void dma_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#endif

// WPS: waves per SIMD the register allocation must allow (workgroups per CU x waves per workgroup / 4)
template <int WGM, int WGN, int STAGES, int WPS>
__global__ void __launch_bounds__(64 * WGM * WGN, WPS) gemm_f64_dma_kernel(GemmP p) {
    constexpr int BM = 128, BN = 128, BK = 16;
    constexpr int NW = WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
    constexpr int LPW = 32 / NW;                       // wave loads (1 KB each) per wave per slab
    static_assert(32 % NW == 0 && STAGES >= 2 && STAGES <= 5, "bad configuration");
    char* smem = (char*)GPMPC_DYN_SMEM();

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
    const int tme = (int)blockIdx.x / p.npad, tne = (int)blockIdx.x % p.npad;
    if (tme >= p.tilesMe || tne >= p.tilesNe) return;

    const int z1 = p.zdiv > 0 ? (int)blockIdx.z % p.zdiv : (int)blockIdx.z;
    const int z2 = p.zdiv > 0 ? (int)blockIdx.z / p.zdiv : 0;
    const double* __restrict__ A = p.A + (long)z1 * p.sA + (long)z2 * p.sA2;
    const double* __restrict__ B = p.B + (long)z1 * p.sB + (long)z2 * p.sB2;
    const dma_rsrc_t rsA = dma_make_rsrc(A, (unsigned)((long)p.M * p.lda * 8));
    const dma_rsrc_t rsB = dma_make_rsrc(B, (unsigned)((long)p.N * p.ldb * 8));
    const int fr = lane & 15, fq = lane >> 4;
    // fragment read offsets inside a slab image: pieces q and 4 + q of the lane's row
    const unsigned sw = (unsigned)((fr >> 1) & 7);
    const unsigned fa0 = (unsigned)((wm * WM + fr) * 128) + (((unsigned)fq ^ sw) << 4), fa1 = fa0 ^ 64u;
    const unsigned fb0 = (unsigned)(16384 + (wn * WN + fr) * 128) + (((unsigned)fq ^ sw) << 4), fb1 = fb0 ^ 64u;

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

        int klo = 0, khi = p.K;
        if (p.kflags & KA_LE_M) khi = min(khi, m0 + BM);
        if (p.kflags & KA_GE_M) klo = max(klo, m0);
        if (p.kflags & KB_LE_N) khi = min(khi, n0 + BN);
        if (p.kflags & KB_GE_N) klo = max(klo, n0);
        klo = klo / BK * BK;
        khi = min(p.K, (khi + BK - 1) / BK * BK);
        const int nk = khi > klo ? (khi - klo) / BK : 0;

        // this lane's source offsets for its LPW pieces of a slab; rows beyond the operand are clamped (their
        // products land in rows / columns that are never stored, the column sums mask them)
        unsigned vo[LPW];
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int j = wave + NW * i;                           // 0..15: A rows 8j.., 16..31: B rows 8(j-16)..
            const int row = 8 * (j & 15) + (lane >> 3);
            const unsigned piece = (unsigned)(lane & 7) ^ (unsigned)((row >> 1) & 7);
            const long grow = (j < 16) ? (long)min(m0 + row, p.M - 1) * p.lda : (long)min(n0 + row, p.N - 1) * p.ldb;
            vo[i] = (unsigned)(grow * 8) + (piece << 4);
        }
        auto request = [&](int t) {                               // slab t of this tile -> ring image t % STAGES
            char* img = smem + (t % STAGES) * DMA_SLAB_BYTES;
            const unsigned so = (unsigned)(klo + t * BK) * 8u;
#pragma unroll
            for (int i = 0; i < LPW; ++i) {
                const int j = wave + NW * i;
                dma_load16(j < 16 ? rsA : rsB, img + 1024 * j, vo[i], so);
            }
        };

        d4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

        auto multiply = [&](int t) {
            const char* img = smem + (t % STAGES) * DMA_SLAB_BYTES;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double2 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const double2*>(img + (h ? fa1 : fa0) + i * 2048);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const double2*>(img + (h ? fb1 : fb0) + j * 2048);
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

        // prologue: STAGES - 1 slabs requested (requests past the end are skipped: the tail waits for everything)
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            if (t < nk) request(t);
        int kt = 0;
        for (; kt + STAGES - 1 < nk; ++kt) {                      // steady state: slab kt + STAGES - 1 exists
            dma_wait<LPW * (STAGES - 2)>();                       // slab kt has landed (this wave's pieces)
            dma_barrier();                                        // ... everybody's, and image (kt - 1) is free
            request(kt + STAGES - 1);
            multiply(kt);
        }
        for (; kt < nk; ++kt) {
            dma_wait<0>();
            dma_barrier();
            multiply(kt);
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
        __syncthreads();  // LDS (red) is reused by the next pass
    }
}

// true if gemm_f64_dma_kernel can run this product
inline bool gemm_dma_supported(const GemmP& p) {
    return !p.a_mc && !p.b_nc && !p.wait_flag && !p.skip00 && !p.done_flags && !p.remap && p.K % 16 == 0 &&
           p.lda % 2 == 0 && p.ldb % 2 == 0 && ((unsigned long)p.A & 15) == 0 && ((unsigned long)p.B & 15) == 0 &&
           (p.sA % 2 == 0) && (p.sB % 2 == 0) && (p.sA2 % 2 == 0) && (p.sB2 % 2 == 0) &&
           (long)p.M * p.lda * 8 < (1L << 32) && (long)p.N * p.ldb * 8 < (1L << 32);
}

template <int WGM, int WGN, int STAGES, int WPS>
inline void launch_gemm_dma(GemmP p, int batch, hipStream_t stream, int resident, int min_pair_blocks = 512) {
    constexpr int BM = 128, BN = 128;
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
    p.npad = p.tilesNe;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_dma_kernel<WGM, WGN, STAGES, WPS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * DMA_SLAB_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f64_dma_kernel<WGM, WGN, STAGES, WPS>), dim3(p.tilesMe * p.npad, 1, batch), dim3(64 * WGM * WGN),
                       STAGES * DMA_SLAB_BYTES, stream, p);
}

// Returns the tile edge used (the caller of EPI_COLSUMSQ sizes `part` with it).
inline int launch_gemm(const GemmP& p, int batch, hipStream_t stream, int force_tile = 0) {
    const int tile = force_tile ? force_tile : gemm_pick_tile(p, batch);
    static const bool use_dma = !(getenv("GPMPC_GEMM_DMA") && atoi(getenv("GPMPC_GEMM_DMA")) == 0);
    if (tile == 128 && use_dma && gemm_dma_supported(p)) {
        launch_gemm_dma<2, 4, 2, 4>(p, batch, stream, 512);
    } else if (tile == 128) {
        launch_gemm_cfg<128, 128, 16, 2, 4>(p, batch, stream, 512);
    } else if (tile == 64) {
        launch_gemm_cfg<64, 64, 16, 2, 2>(p, batch, stream, 1024);
    } else if (p.K % 32 == 0) {
        launch_gemm_cfg<32, 32, 32, 2, 2>(p, batch, stream, 1024);
    } else {
        launch_gemm_cfg<32, 32, 16, 2, 2>(p, batch, stream, 1024);
    }
    return tile;
}

}  // namespace gpmpc
