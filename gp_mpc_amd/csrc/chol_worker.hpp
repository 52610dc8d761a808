// Tile-owner workers of the blocked Cholesky: the bulk of the factorisation (panel rows and trailing
// updates) as ONE persistent kernel next to the chain kernel (chol_chain.hpp), instead of two GEMM
// launches per panel step.
//
// The lower triangle is cut into 64 x 64 tiles (i, j), i >= j; the tiles with j >= 1 are numbered column by
// column, tile t belongs to worker t mod NW and LIVES IN THAT WORKER'S REGISTERS until its last update (at most
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
// With a courier (template argument; the default since r03) step 2 is not the owners' any more: the last workgroup of
// the launch owns no tiles, receives the three tiles of row k+3 from their owners behind the update of step k (they come
// first in part 3; stored to K, counted in handed[k+1]) and makes L(k+2,k) and the two hand-off tiles as the chain's
// publications appear -- see the courier's loop below.  Part 3 also looks one panel ahead: the tiles of column k+1 are
// updated first and turned into L(i,k+1) as soon as the chain has published inv_{k+1} (no blocking wait).
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
#include "lds_dma.hpp"

namespace gpmpc {

constexpr int WORKER_MAXT = 9;            // 9 tiles x 224 workers hold the 2015 tiles of Np = 4096
constexpr int WORKER_MAXT_COURIER = 10;   // ... with one workgroup as courier it is 10 x 223 (the larger kernel is slower per tile)
constexpr int WORKER_THREADS = 512;
// LDS: two operand pairs of the trailing update (2 x 2 x 32 KB, DMA images); the two padded 64 x 65 blocks of the
// panel / hand-off products alias the first of them; then the slot table.  > 80 KB also keeps one worker per CU.
constexpr int WORKER_PAIR_BYTES = 2 * 64 * 64 * 8;
constexpr int WORKER_LDS_BYTES = 2 * WORKER_PAIR_BYTES + 256;

// Global addresses are a workgroup-uniform base plus ONE per-thread unsigned 32-bit BYTE offset, so that the
// accesses can take the SGPR-base form and the addresses cost one register.
__device__ __forceinline__ const double& at_byte(const double* base, unsigned byte_off) {
    return *(const double*)((const char*)base + byte_off);
}
__device__ __forceinline__ double& at_byte(double* base, unsigned byte_off) { return *(double*)((char*)base + byte_off); }

// A store that another workgroup will read behind a flag: write-through (an agent-scope relaxed atomic store is a
// global_store_dwordx2 sc1) when `wt`, so that the publication needs no L2 write-back (wg_sync.hpp), else plain.
__device__ __forceinline__ void st_pub(double* base, unsigned byte_off, double v, int wt) {
#ifdef GPMPC_EMULATED
    at_byte(base, byte_off) = v;
#else
    if (wt) __hip_atomic_store(&at_byte(base, byte_off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else at_byte(base, byte_off) = v;
#endif
}
#define WORKER_RELEASE() do { if (!wt) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); } while (0)

template <int MAXT, bool COURIER>
__global__ void __launch_bounds__(WORKER_THREADS) chol_worker_kernel(double* Kmat, double* L, const double* Inv, long ld,
                                                                     long sBatch, int nb_all, int* flags, long sFlags,
                                                                     int crow_mode, int spin_limit, int kb, int ksteps,
                                                                     int* ready, long long* trace, int lookahead, int wt) {
    // optional time stamps (100 MHz wall clock) for tools/worker_trace.py: [launch][worker][step][4] from entry 4096 on
#ifdef GPMPC_EMULATED
#define WORKER_STAMP(i) ((void)0)
#define COURIER_STAMP(i) ((void)0)
#else
#define COURIER_STAMP(i) do { if (trace && threadIdx.x == 0) trace[300000 + (kb + k) * 8 + (i)] = wall_clock64(); } while (0)
#define WORKER_STAMP(i) do { if (trace && threadIdx.x == 0) trace[4096 + ((((kb ? 1L : 0L) * 256 + blockIdx.x) * 64 + (kb + k)) * 4) + (i)] = wall_clock64(); } while (0)
#endif
    double* smem = GPMPC_DYN_SMEM();
    int* slot = (int*)((char*)smem + 2 * WORKER_PAIR_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;                 // this wave's 16 x 32 piece: rows 16 wr, columns 32 wc
    // With `courier` the LAST workgroup of the launch is not a tile owner but the courier of the chain (see below); the
    // others are the NW regular workers.
    constexpr bool has_courier = COURIER;              // (the launch has >= 2 workgroups then: host)
    const int w = blockIdx.x, NW = has_courier ? (int)gridDim.x - 1 : (int)gridDim.x;
    const bool is_courier = has_courier && w == NW;
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
    int* handed = fl + chain_handed_index(nb_all) + kb;
    if (tid == 0) flag_store(progress, 1);
    if (ready && tid == 0) {                       // "all workgroups of this launch are resident" for the host's gates
        int* rd = ready + (long)blockIdx.z * sFlags;
        if (__hip_atomic_fetch_add(rd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == (int)gridDim.x) flag_store(rd + 1, 1);
    }
#ifndef GPMPC_EMULATED
    if (tid == 0) flag_store(progress + 256, (int)((wall_clock64() / 100) & 0x3fffffff));   // start time, us
#endif
    const int ntiles = (nb - 1) * nb / 2 - 1;    // resident tiles: (i, j), 1 <= j <= i, without (1,1)
    // per-thread BYTE offset inside a 64 x 64 tile of an [ld]-strided matrix: csub = the thread's first
    // accumulator element (sub-tile row crow(lane, 0), column lane & 15)
    // (crow(lane, r) is linear in r with a uniform step -- 4 rows in the gfx950 f64 map -- so element r sits
    //  r * cstep rows below element 0: one offset register, the step goes into the uniform base pointer)
    const long cstep = (long)(crow(0, 1, crow_mode) - crow(0, 0, crow_mode)) * ld;
    const unsigned csub = 8u * (unsigned)((16 * wr + crow(lane, 0, crow_mode)) * (int)ld + 32 * wc + (lane & 15));
    // every product takes its operands by DMA (lds_dma.hpp): a 64 x 64 block becomes four 16-column slab images of
    // [64 rows][128 B], 16-byte piece c of row r stored at position c ^ ((r >> 1) & 7) (the layout of
    // gemm_f64_dma.hpp: one conflict-free ds_read_b128 per fragment and K pair).  Wave v < 4 loads rows 8v .. 8v+7 and
    // 8v+32 .. 8v+39 of every slab: dvo = this lane's source byte offset inside the block, fa / fb = fragment read offsets.
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const int drow = 8 * (wave & 3) + (lane >> 3);
    const unsigned dvo = 8u * (unsigned)(drow * (int)ld) + ((unsigned)((lane & 7) ^ ((drow >> 1) & 7)) << 4);
    const unsigned fsw = (unsigned)(((lane & 15) >> 1) & 7), fq = (unsigned)(lane >> 4);
    const unsigned fa0 = (unsigned)((16 * wr + (lane & 15)) * 128) + ((fq ^ fsw) << 4), fa1 = fa0 ^ 64u;
    const unsigned fb0 = (unsigned)(32768 + (32 * wc + (lane & 15)) * 128) + ((fq ^ fsw) << 4), fb1 = fb0 ^ 64u;

    // my tiles: (ti[n], tj[n]) in LDS (workgroup-uniform; ti < 0 = none / finished), the tiles themselves in
    // registers C[n].  The slot index is a run-time value: the code that touches a tile is
    // addressed through a switch over compile-time indices so that C[] never becomes an indexed (scratch) array.
    int* ti = slot + 4;
    int* tj = ti + MAXT;
    d4 C[MAXT][2];
    if (tid < MAXT) {
        const int t = w + tid * NW;
        int i = -1, j = 0;
        if (t < ntiles && !is_courier) {
            // t + 1 enumerates the triangle (r, c) = (i-1, j-1), c <= r, COLUMN by column (column c holds rows c .. nb-2).
            // The tiles live at step k are those of the columns > k -- a suffix of this numbering -- so dealing them round
            // robin leaves every worker ceil(live / NW) of them at EVERY step.  (r01-r02 numbered row by row: the same
            // total per worker, but at step 15 the busiest worker had 8 live tiles against a mean of 5.2, at step 24 7
            // against 3.5 -- over the first 32 steps 249 tile updates on the critical path instead of 186.)
            int c = 0, m = t + 1;
            while (m >= nb - 1 - c) { m -= nb - 1 - c; ++c; }
            i = c + m + 1;
            j = c + 1;
            if (has_courier && i == 2) i = -1;        // row 2 is the courier's first row: it takes (2,1), (2,2) from K
        }
        ti[tid] = i; tj[tid] = j;
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < MAXT; ++n) {
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

    // k-independent building blocks (used by the step loop and by the courier)
    // operand pair of tile (i, j) for this step: blocks L(i,k) and L(j,k) -> pair image `pp` (0 / 1) by DMA,
    // 8 loads per wave; `update`: c -= L(i,k) L(j,k)^T from a landed pair image
    // Issued by ONE wave per SIMD (waves 0-3, two row groups each): the other four go straight on to their
    // matrix instructions, so the ~150 scalar instructions of a request do not idle the matrix pipes.
    auto request_blocks = [&](const double* pa, const double* pb, int pp, bool with_a, bool with_b = true) {
        if (swave >= 4) return;
        char* img = (char*)smem + pp * WORKER_PAIR_BYTES + 1024 * swave;
        const unsigned half = (unsigned)(32 * ld * 8);        // rows 8 (wave + 4) .. : 32 rows further down
        if (with_a) dma_load_block64(dma_make_rsrc(pa, (unsigned)(64 * ld * 8)), img, dvo, half);
        if (with_b) dma_load_block64(dma_make_rsrc(pb, (unsigned)(64 * ld * 8)), img + 32768, dvo, half);
    };
    // c0, c1 += sgn * A B^T on this wave's 16 x 32 piece, operands from the landed pair image pp
    // (imgA / imgB: block images of the A and the B operand -- any landed 32 KB image serves as either)
    auto product_ab = [&](const char* imgA, const char* imgB, d4& c0, d4& c1, bool negate) {
        const char* img = imgA;
        const long boff = (imgB - imgA) - 32768;       // fb0 / fb1 carry the B part's offset inside a pair image
        // Fully unrolled (r05): the 24 fragment reads of a tile are in flight under its 32 matrix instructions (227 registers,
        // no spill with this compiler; r03 kept the slab loop rolled because it spilled then).  Same-box A/B, C2: slab loop /
        // two slabs / all four unrolled: chain 1.2015 / 1.1670 / 1.1435 ms, step 4.140 / 4.114 / 4.093 ms
        // (profiles/r05_worker_update_unroll_ab.txt) -- the workers bound every quarter of the chain by then.
        // The subtraction is the matrix instruction's own NEG modifier on A (mfma16_nega), not two v_xor per K pair.
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double2 a = *reinterpret_cast<const double2*>(img + 8192 * sl + (h ? fa1 : fa0));
                const double2 b0 = *reinterpret_cast<const double2*>(img + boff + 8192 * sl + (h ? fb1 : fb0));
                const double2 b1 = *reinterpret_cast<const double2*>(img + boff + 8192 * sl + 2048 + (h ? fb1 : fb0));
                if (negate) {
                    c0 = mfma16_nega(a.x, b0.x, c0);
                    c1 = mfma16_nega(a.x, b1.x, c1);
                    c0 = mfma16_nega(a.y, b0.y, c0);
                    c1 = mfma16_nega(a.y, b1.y, c1);
                } else {
                    c0 = mfma16(a.x, b0.x, c0);
                    c1 = mfma16(a.x, b1.x, c1);
                    c0 = mfma16(a.y, b0.y, c0);
                    c1 = mfma16(a.y, b1.y, c1);
                }
            }
    };
    auto product = [&](int pp, d4& c0, d4& c1, bool negate) {
        const char* img = (const char*)smem + pp * WORKER_PAIR_BYTES;
        product_ab(img, img + 32768, c0, c1, negate);
    };
    auto update = [&](int pp, d4& c0, d4& c1) { product(pp, c0, c1, true); };
    // a resident tile (this wave's two accumulators) as the A operand of pair image 0, in the slab layout
    auto tile_to_image = [&](const d4& c0, const d4& c1, int img_off = 0) {
        char* img = (char*)smem + img_off;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wr + crow(lane, r, crow_mode), cc = lane & 15;
            const unsigned o = (unsigned)(row * 128) + ((unsigned)((cc >> 1) ^ ((row >> 1) & 7)) << 4) + 8u * (cc & 1);
            *reinterpret_cast<double*>(img + 8192 * (2 * wc) + o) = c0[r];
            *reinterpret_cast<double*>(img + 8192 * (2 * wc + 1) + o) = c1[r];
        }
    };

    // ---- the courier (last workgroup of the launch): row k + 2 of every panel step k ------------------------------------
    // The chain needs, after its NEXT leaf, the tiles (k+2,k+1) and (k+2,k+2) with the update of column k; that update
    // needs L(k+2,k) = A(k+2,k) inv_kk^T.  Tile owners reach those three products one flag hop apart (panel tile's owner
    // -> hand-off tiles' owners), ~9 us after the chain's publication even when they stand waiting for it -- a microsecond
    // after the chain looks for the tiles (8 us, two thirds into its next leaf), so that it fetches them after the leaf
    // instead of behind its last panel (+1.3 us of load latency) and often waits for them as well.  The courier does
    // nothing else: the owners hand it the three tiles of row k + 2 one step EARLY (with the updates of columns < k,
    // stored to K behind their update of step k - 1; handed[k] counts them), it spins on the chain's publications and runs
    // the products as their operands appear: L(k+2,k) and the diagonal tile behind inv_kk, the other tile behind the panel row.
    if (is_courier) {
        d4 c1[2], c2[2];
        for (int k = 0; k + 2 < nb && k < ksteps; ++k) {
            const int i = k + 2;
            if (tid == 0) flag_store(progress, 1 + 4 * k);
            // the three tiles with the updates of columns < k: at k = 0 they are in K (K build / previous launch).  They arrive
            // with or after inv_kk (time stamps, tools/courier_trace.py), so both are awaited together and everything is
            // requested at once -- a flag poll BEHIND the tile loads would wait for them (in-order return), 3-4 us.
            if (!wg_wait2(&leafdone[k], 1, k > 0 ? &handed[k] : nullptr, 3, err, spin_limit, slot, 6000000 + 1000 * k)) return;
            COURIER_STAMP(0);
            const double* s1 = Kb + (long)(64 * i) * ld + 64 * (k + 1);
            const double* s2 = Kb + (long)(64 * i) * ld + 64 * i;
            // stage 1: L(i,k) like every other tile of the panel column -- the workers' step k waits for the whole column
            // (colready) -- and the diagonal tile
            request_blocks(Kb + (long)(64 * i) * ld + 64 * k, Ib + (long)(64 * k) * ld + 64 * k, 0, true);   // A(i,k), inv_kk -> image 0
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c1[0][r] = at_byte(s1 + r * cstep, csub); c1[1][r] = at_byte(s1 + r * cstep, csub + 128u);
                c2[0][r] = at_byte(s2 + r * cstep, csub); c2[1][r] = at_byte(s2 + r * cstep, csub + 128u);
            }
            COURIER_STAMP(1);
            dma_wait<0>();
            __syncthreads();
            d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
            product(0, acc[0], acc[1], false);                                                     // L(i,k)
            double* dl = Lb + (long)(64 * i) * ld + 64 * k;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st_pub(dl + r * cstep, csub, acc[0][r], wt);
                st_pub(dl + r * cstep, csub + 128u, acc[1][r], wt);
            }
            GPMPC_DRAIN_VM();                                                                      // L(i,k) is out: the column count
            tile_to_image(acc[0], acc[1], WORKER_PAIR_BYTES);                                      // ... -> image 1, A part
            __syncthreads();
            if (tid == 0) {
                WORKER_RELEASE();
                GPMPC_DRAIN_VM();
                flag_store(&row2done[k], 1);
                const int before = __hip_atomic_fetch_add(&pancount[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (before + 1 == nb - k - 2) flag_store(&colready[k], 1);
            }
            const char* img1 = (const char*)smem + WORKER_PAIR_BYTES;
            product_ab(img1, img1, c2[0], c2[1], true);                                            // (i,i)   -= L(i,k) L(i,k)^T
            double* d1 = Kb + (long)(64 * i) * ld + 64 * (k + 1);
            double* d2 = Kb + (long)(64 * i) * ld + 64 * i;
            COURIER_STAMP(2);                       // (its stores wait until the end: a flag poll behind stores waits for them)
            // stage 2, behind the chain's panel row: the off-diagonal tile, then both go to the chain
            if (!wg_wait2(&pan1[k], 1, nullptr, 0, err, spin_limit, slot, 6600000 + 1000 * k)) return;
            COURIER_STAMP(3);
            request_blocks(nullptr, Lb + (long)(64 * (k + 1)) * ld + 64 * k, 1, false);            // L(k+1,k) -> image 1, B part
            dma_wait<0>();
            __syncthreads();
            product_ab(img1, img1 + 32768, c1[0], c1[1], true);                                    // (i,k+1) -= L(i,k) L(k+1,k)^T
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st_pub(d1 + r * cstep, csub, c1[0][r], wt); st_pub(d1 + r * cstep, csub + 128u, c1[1][r], wt);
                st_pub(d2 + r * cstep, csub, c2[0][r], wt); st_pub(d2 + r * cstep, csub + 128u, c2[1][r], wt);
            }
            GPMPC_DRAIN_VM();
            __syncthreads();
            if (tid == 0) {
                WORKER_RELEASE();
                GPMPC_DRAIN_VM();
                flag_store(&tdone[2 * k], 1);
                flag_store(&tdone[2 * k + 1], 1);
            }
            COURIER_STAMP(4);
            __syncthreads();
        }
        return;
    }

    // One panel step = parts 1, 2, 3 below.  Part 3 may leave its update pipeline once ("look-ahead": the panel tiles of
    // column k + 1 become L(i,k+1) as soon as inv_{k+1} is out) -- that is a second pass through part 1 with
    // `ahead` = 1, written as a state of this ONE loop so that every heavy block (panel product, update pipeline)
    // exists once in the kernel (two copies of the panel product pushed the kernel from 197 to 256 VGPRs + 159 spills).
    int k = 0, ahead = 0;                            // ahead: bit 0 = early panel tiles are due, bit 1 = early hand-off tiles
    // part 3: coordinates of the resident tiles as the step began, in LDS behind ti / tj (never rewritten during the step).
    // (r01-r04a kept them in two int arrays read with a run-time slot index: the compiler put those into scratch, and every
    // tile of the update began, in the waves that request the next operand pair, with two scratch loads and a wait in front of
    // the DMA requests -- on the one wave per SIMD whose matrix instructions nobody else can take.)
    int* lis = tj + MAXT;
    int* ljs = lis + MAXT;
    unsigned live = 0, urgent = 0, hand = 0, todo = 0;   //     masks: live this step / column k + 1 / next hand-off tiles / not yet updated
    bool early = false, earlyh = false;              //         tiles of column k + 1 wait for inv_{k+1} / hand-off tiles for row k + 3
    bool give = false;                               //         this step hands the tiles of row k + 3 to the courier
    int givepend = 0;                                //         tiles stored for the courier whose count is not yet published
    int peekv = 0, cur = -1, pp = 0;                 //         bit 0: leafdone[k+1], bit 1: row2done[k+1] and pan1[k+1], as last seen
                                                     //         by thread 0; pipeline state
    for (; k + 2 < nb && k < ksteps;) {
        if (!ahead) WORKER_STAMP(0);
        auto request = [&](int i, int j, int pp) {
            request_blocks(Lb + (long)(64 * i) * ld + 64 * k, Lb + (long)(64 * j) * ld + 64 * k, pp, true);
        };
        // ---- 1. panel tiles of column k.  Column 0 never receives an update, so its tiles are not kept in
        //         registers: at k = 0 worker w takes rows 2 + w, 2 + w + NW, ... straight from K.
        for (int i = (has_courier ? 3 : 2) + w; k == 0 && !ahead && i < nb; i += NW) {      // (row 2: the courier's)
            if (tid == 0) flag_store(progress, 1 + 4 * k + 1);
            if (!wg_wait2(&leafdone[0], 1, nullptr, 0, err, spin_limit, slot, 2000000 + w)) return;
            request_blocks(Kb + (long)(64 * i) * ld, Ib, 0, true);                  // A(i,0), inv_00
            dma_wait<0>();
            __syncthreads();
            d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
            product(0, acc[0], acc[1], false);
            double* dst = Lb + (long)(64 * i) * ld;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st_pub(dst + r * cstep, csub, acc[0][r], wt);
                st_pub(dst + r * cstep, csub + 128u, acc[1][r], wt);
            }
            GPMPC_DRAIN_VM();
            __syncthreads();
            if (tid == 0) {
                WORKER_RELEASE();
                GPMPC_DRAIN_VM();
                if (i == 2) flag_store(&row2done[0], 1);
                const int before = __hip_atomic_fetch_add(&pancount[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (before + 1 == nb - 2) flag_store(&colready[0], 1);
            }
            __syncthreads();
        }
        const int kk = k + (ahead ? 1 : 0);                // the step whose panel / hand-off tiles are due
        if (!ahead || (ahead & 1)) {
#pragma unroll
            for (int n = 0; n < MAXT; ++n) {
                const int i = __builtin_amdgcn_readfirstlane(ti[n]), j = __builtin_amdgcn_readfirstlane(tj[n]);
                if (i < 0 || j != kk) continue;            // (workgroup-uniform)
                if (tid == 0) flag_store(progress, 1 + 4 * kk + 1);
                if (!wg_wait2(&leafdone[kk], 1, nullptr, 0, err, spin_limit, slot, 2000000 + 1000 * kk + w)) return;
                request_blocks(nullptr, Ib + (long)(64 * kk) * ld + 64 * kk, 0, false);   // inv_kk (zeros above the diagonal)
                tile_to_image(C[n][0], C[n][1]);
                dma_wait<0>();
                __syncthreads();                           // (also waits for the LDS stores of tile_to_image)
                d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
                product(0, acc[0], acc[1], false);
                double* dst = Lb + (long)(64 * i) * ld + 64 * kk;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st_pub(dst + r * cstep, csub, acc[0][r], wt);
                    st_pub(dst + r * cstep, csub + 128u, acc[1][r], wt);
                }
                GPMPC_DRAIN_VM();
                __syncthreads();                           // (also: everybody is done with A and B)
                if (tid == 0) {
                    WORKER_RELEASE();
                    GPMPC_DRAIN_VM();
                    if (i == kk + 2) flag_store(&row2done[kk], 1);
                    // count the tile; whoever completes the column raises the flag the consumers poll
                    const int before = __hip_atomic_fetch_add(&pancount[kk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (before + 1 == nb - kk - 2) flag_store(&colready[kk], 1);
                    ti[n] = -1;
                }
                __syncthreads();
            }
        }
        // ---- 2. (kk+2,kk+1), (kk+2,kk+2) for the chain (with a courier: its job, the tiles were handed to it a step ago)
        if (!has_courier && (!ahead || (ahead & 2))) {
#pragma unroll
            for (int n = 0; n < MAXT; ++n) {
                const int i = __builtin_amdgcn_readfirstlane(ti[n]), j = __builtin_amdgcn_readfirstlane(tj[n]);
                if (i != kk + 2 || j <= kk) continue;      // j is kk+1 or kk+2
                if (tid == 0) flag_store(progress, 1 + 4 * kk + 2);
                if (!wg_wait2(&row2done[kk], 1, j == kk + 1 ? &pan1[kk] : nullptr, 1, err, spin_limit, slot, 3000000 + 1000 * kk + w))
                    return;
                request_blocks(Lb + (long)(64 * (kk + 2)) * ld + 64 * kk, Lb + (long)(64 * j) * ld + 64 * kk, 0, true);
                dma_wait<0>();
                __syncthreads();
                double* dst = Kb + (long)(64 * (kk + 2)) * ld + 64 * j;
                update(0, C[n][0], C[n][1]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st_pub(dst + r * cstep, csub, C[n][0][r], wt);
                    st_pub(dst + r * cstep, csub + 128u, C[n][1][r], wt);
                }
                if (wt) wg_publish_wt(&tdone[2 * kk + (j == kk + 2 ? 1 : 0)], 1); else wg_publish(&tdone[2 * kk + (j == kk + 2 ? 1 : 0)], 1);
                if (tid == 0) ti[n] = -1;
                __syncthreads();                           // A, B free again; ti[] visible
            }
        }
        auto coords = [&](int m, int& ci, int& cj) {   // coordinates of slot m for a run-time m: two broadcast LDS reads
            ci = __builtin_amdgcn_readfirstlane(lis[m]);
            cj = __builtin_amdgcn_readfirstlane(ljs[m]);
        };
        // order of part 3: panel tiles of the next step, then its hand-off tiles, then by slot
        auto pick = [&](unsigned t) {
            if (t == 0) return -1;
            if (has_courier) return (int)__builtin_ctz((t & hand) ? (t & hand) : (t & urgent) ? (t & urgent) : t);
            return (int)__builtin_ctz((t & urgent) ? (t & urgent) : (t & hand) ? (t & hand) : t);
        };
        if (ahead) {
            // back from the early products: the rest of this step's updates, pipeline restarted
            if (ahead & 1) early = false;
            if (ahead & 2) earlyh = false;
            ahead = 0;
            if (cur >= 0) {
                int ci, cj;
                coords(cur, ci, cj);
                request(ci, cj, 0);
                pp = 0;
            }
        } else {
        // ---- 3. all other live tiles, software-pipelined: operands of the next tile travel global -> registers
        //         while the MFMAs of the current one run from LDS
        WORKER_STAMP(1);
        // live tiles of this step, once: coordinates in scalar registers, membership as a bit mask (looking the next
        // tile up in the LDS table between two tiles cost ~0.7 us per tile: 16 dependent LDS round trips)
        // Look-ahead: the tiles of column k + 1 are updated FIRST; as soon as the chain has published inv_{k+1} (peeked at
        // once per tile, the load travels behind the matrix instructions) they become L(i,k+1) right here instead of at
        // the top of step k + 1, where every worker would stand waiting for the slowest one's product (measured before:
        // 9-11 us of a 38 us step in the first quarter, tools/worker_trace.py).  Never a blocking wait: if the
        // publication does not come while this worker still updates, part 1 of the next step does it as before.
        // The two tiles the chain needs after the NEXT leaf, (k+3,k+2) and (k+3,k+3), come second and are handed over as
        // soon as their operands L(k+3,k+1) [row2done] and L(k+2,k+1) [pan1] exist -- otherwise their owner would reach
        // them only after its whole part 3, and the chain, one step ahead thanks to the early panel products, would
        // stand waiting for it every other step (seen in the time stamps: 0.6 / 16 / 0.6 / 22 us).
        live = 0; urgent = 0; hand = 0;
        const bool next_here = k + 1 < ksteps && k + 3 < nb;           // the next step belongs to this launch and has panel tiles
        const bool look = lookahead && next_here;
        // with a courier: the three tiles of row k + 3 go to it behind this step's update (stored to K, handed[k+1]);
        // they come first, no product of step k + 1 is this worker's any more
        give = has_courier && next_here;
        if (tid < MAXT) { lis[tid] = ti[tid]; ljs[tid] = tj[tid]; }   // (read behind the barriers of the wait below)
#pragma unroll
        for (int n = 0; n < MAXT; ++n) {
            const int in = __builtin_amdgcn_readfirstlane(ti[n]), jn = __builtin_amdgcn_readfirstlane(tj[n]);
            if (in >= 0 && jn > k) {
                live |= 1u << n;
                if (give && in == k + 3) hand |= 1u << n;
                else if (look && jn == k + 1) urgent |= 1u << n;
                else if (look && !has_courier && in == k + 3 && jn >= k + 2) hand |= 1u << n;
            }
        }
        if (live == 0) { ++k; continue; }
        if (tid == 0) flag_store(progress, 1 + 4 * k + 3);
        if (!wg_wait2(&colready[k], 1, &pan1[k], 1, err, spin_limit, slot, 4000000 + 1000 * k + w)) return;
        WORKER_STAMP(2);
        todo = live;                                   // tiles still to update; order: the urgent ones, then by slot
        early = urgent != 0;
        earlyh = hand != 0 && !has_courier;
        peekv = 0;
        givepend = 0;
        cur = pick(todo);
        {
            int ci, cj;
            coords(cur, ci, cj);
            request(ci, cj, 0);
        }
        pp = 0;
        }   // (!ahead)
        bool stop = false;                             // leave the pipeline for the early panel product
#pragma unroll 1
        while (cur >= 0 && !stop) {
#pragma unroll
            for (int n = 0; n < MAXT; ++n) {
                if (n != cur || stop) continue;        // (slots before the first / between live tiles)
                todo &= ~(1u << n);
                const int nxt = pick(todo);
                // an early product may start once its tiles carry this step's update and its operands are out
                const bool any_early = early || earlyh;
                const int ready = ((early && (todo & urgent) == 0) ? 1 : 0) | ((earlyh && (todo & hand) == 0) ? 2 : 0);
                if (any_early && tid == 0) slot[1 + pp] = peekv & ready;   // (two words in turn: no barrier between a read
                dma_wait<0>();                         // this wave's pieces of the current pair have landed   and the next write)
                if (any_early) __syncthreads();        // (orders the word above as well)
                else dma_barrier();                    // ... everybody's; the other pair image is free
                if (any_early && ready != 0) {
                    const int go = slot[1 + pp];
                    if (go != 0) { stop = true; ahead = go; }
                }
                if (givepend) {                        // the stores of the tiles given away have drained (dma_wait<0> above
                    if (tid == 0) {                    // is vmcnt(0)) in every wave: barrier passed -> publish the count
                        WORKER_RELEASE();
                        GPMPC_DRAIN_VM();
                        __hip_atomic_fetch_add(&handed[k + 1], givepend, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    givepend = 0;
                }
                if (nxt >= 0 && !stop) {
                    int ci, cj;
                    coords(nxt, ci, cj);
                    request(ci, cj, pp ^ 1);
                }
                update(pp, C[n][0], C[n][1]);
                if (give && ((hand >> n) & 1u)) {      // a tile of row k + 3: to K for the courier, retired here
                    int gi, gj;
                    coords(n, gi, gj);
                    double* dst = Kb + (long)(64 * gi) * ld + 64 * gj;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        st_pub(dst + r * cstep, csub, C[n][0][r], wt);
                        st_pub(dst + r * cstep, csub + 128u, C[n][1][r], wt);
                    }
                    if (tid == 0) ti[n] = -1;
                    ++givepend;
                }
                if (any_early && tid == 0) {           // consumed at the next tile: the latency hides behind it
                    peekv = 0;
                    if (early && flag_load(&leafdone[k + 1]) >= 1) peekv |= 1;
                    if (earlyh && flag_load(&row2done[k + 1]) >= 1 && flag_load(&pan1[k + 1]) >= 1) peekv |= 2;
                }
                pp ^= 1;
                cur = nxt;
#ifndef GPMPC_EMULATED
                if (trace && threadIdx.x == 0 && blockIdx.x < 8 && kb == 0)      // per-tile stamps of the first 8 workers
                    trace[135168 + ((long)blockIdx.x * 64 + k) * 10 + n] = wall_clock64();
#endif
            }
        }
        if (givepend) GPMPC_DRAIN_VM();                // (a tile given away by the last iteration: its stores)
        __syncthreads();                               // the pair images alias A and B of the next step / of the early product
        if (givepend) {
            if (tid == 0) {
                WORKER_RELEASE();
                GPMPC_DRAIN_VM();
                __hip_atomic_fetch_add(&handed[k + 1], givepend, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            givepend = 0;
        }
        if (stop) continue;                            // pipeline drained (nothing requested): parts 1 / 2 of step k + 1 as `ahead` says, then back
        WORKER_STAMP(3);
        ++k;
    }
#undef WORKER_STAMP
    // a launch that stops before the last step hands its live tiles back through K: the next launch (from block
    // kb + ksteps on, with fewer workers -- the freed CUs take the inverse pipeline) reloads them
    if (k + 2 < nb) {
#pragma unroll
        for (int n = 0; n < MAXT; ++n) {
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
