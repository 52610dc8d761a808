// Persistent chain kernel of the blocked Cholesky: ONE workgroup per matrix owns a CU for the whole
// factorisation and runs the sequential part of every panel step back to back,
//
//     leaf(k):  L_kk = chol(A_kk), inv_kk = L_kk^-1                    (leaf_body, LDS resident)
//     tile (k+1,k):    L_{k+1,k} = A_{k+1,k} inv_kk^T                   (MFMA on LDS operands)
//     tile (k+1,k+1):  A_{k+1,k+1} -= L_{k+1,k} L_{k+1,k}^T             (stays in LDS: next leaf's input)
//
// while the bulk of each step -- the panel rows below and the trailing update -- runs either in ordinary GEMM
// launches on a second stream or in the persistent tile-owner workers (chol_worker.hpp).  The two sides meet
// through global flags (wg_sync.hpp):
//     leafdone[k]  chain -> panel(k) workgroups        (L_kk^-1 is in memory)
//     pan1[k]      chain -> trailing(k) workgroups     (L_{k+1,k} is in memory)
//     tdone[k][2]  trailing(k) -> chain                (tiles (k+2,k+1) and (k+2,k+2) carry step k's update)
// Why: measured on MI355X, the leaf slows down 3-8x as soon as it shares a CU with MFMA-heavy waves
// (tools/ubench/contention_bench.hip) and CU-masked streams do not isolate it, so overlapping the chain
// with the bulk work needs a workgroup that keeps a CU to itself: this kernel asks for ~150 KB of LDS,
// which no other workgroup of this library fits beside.
// grid (1, 1, batch), 256 threads, dynamic LDS CHAIN_LDS_BYTES.
#pragma once
#include "leaf64.hpp"
#include "wg_sync.hpp"

namespace gpmpc {

constexpr int CHAIN_LDS_BYTES = 152576;   // S, T, U, P (64 x LS each), the packed lower triangle Qp, Dr, slot, the store table
constexpr int CHAIN_TRI_PAIRS = 1056;     // 16-byte column pairs (r, 2c), 2c <= r, of a 64 x 64 lower triangle

// flag layout per matrix (ints): [0] error, [1 .. nb] leafdone, [1+nb .. 2nb] pan1, [1+2nb .. 1+4nb) tdone[k][2],
// and for the tile-owner workers (chol_worker.hpp): [1+4nb .. 1+5nb) pancount, [1+5nb .. 1+6nb) row2done,
// [1+6nb .. 1+7nb) colready, [1+7nb .. 1+7nb+512) progress / start time of worker w (diagnostics),
// [1+7nb+512 .. +8) arrival counter and "all resident" flag of the second, third and fourth worker launch
// [1+7nb+512+8 .. +nb) handed[k]: how many of the three tiles of row k+2 the workers have handed to the courier
__host__ __device__ inline int chain_flag_count(int nb) { return 1 + 8 * nb + 512 + 8; }
__host__ __device__ inline int chain_handed_index(int nb) { return 1 + 7 * nb + 512 + 8; }
__host__ __device__ inline int chain_colready_index(int nb, int k) { return 1 + 6 * nb + k; }
__host__ __device__ inline int chain_pan1_index(int nb, int k) { return 1 + nb + k; }
__host__ __device__ inline int chain_ready_index(int nb) { return 1 + 7 * nb + 512; }   // counter; the flag is the next word

// Gate for the side queue: one tiny workgroup per matrix that returns once the chain kernel has
// published its first leaf, i.e. once every chain workgroup is resident.  Without it a batch of bulk
// workgroups could fill the LDS of all CUs while polling and keep the chain from ever being placed.
__global__ void __launch_bounds__(64) chain_gate_kernel(int* flags, long sFlags, int spin_limit) {
    __shared__ int slot;
    int* fl = flags + (long)blockIdx.x * sFlags;
    wg_wait2(fl + 1, 1, nullptr, 0, fl, spin_limit, &slot);
}

// Generic gate for a queue: returns once *f0 >= v0 (and *f1 >= v1 if f1 is given), per matrix of the batch.
__global__ void __launch_bounds__(64) flag_gate_kernel(int* flags, long sFlags, int i0, int v0, int i1, int v1,
                                                       int spin_limit) {
    __shared__ int slot;
    int* fl = flags + (long)blockIdx.x * sFlags;
    wg_wait2(fl + i0, v0, i1 >= 0 ? fl + i1 : nullptr, v1, fl, spin_limit, &slot, 5000000 + i0);
}

// Tiles of A(k+1,k+1) -= U U^T (U = L(k+1,k), 64 x 64 in LDS, row stride LS) into S.  part 0: tiles (i, 0), i = `who`
// (0..3); part 1: the six tiles (i, j), 1 <= j <= i <= 3, dealt to `who` = 0..2 (two each).  The block before the
// update is read from S itself or -- from_q -- from the packed lower triangle Qp the prefetch filled ((r, c) at
// r (r + 1) / 2 + c; entries above the diagonal do not exist there and are not needed: nothing reads S above it).
__device__ __forceinline__ void chain_diag_tiles(double* S, const double* U, const double* Qp, bool from_q, int who, int nwho,
                                                 int part, int lane, int crow_mode) {
    auto subtract = [&](int i, int j, const d4& acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = 16 * i + crow(lane, r, crow_mode), cc = 16 * j + (lane & 15);
            const double a = from_q ? (cc <= rr ? Qp[rr * (rr + 1) / 2 + cc] : 0.0) : S[rr * LS + cc];
            S[rr * LS + cc] = a - acc[r];
        }
    };
    // this caller's tiles: at most two (part 1: six tiles over three waves); their products run interleaved
    int cnt = 0, mine = 0, ti[2] = {0, 0}, tj[2] = {0, 0};
    for (int i = part; i < 4; ++i)
        for (int j = part; j <= (part ? i : 0); ++j, ++cnt)
            if (cnt % nwho == who && mine < 2) { ti[mine] = i; tj[mine] = j; ++mine; }
    if (mine == 2) {
        d4 a0 = d4{0.0, 0.0, 0.0, 0.0}, a1 = a0;
        lds_mm16k_x2<true, 64>(U, 16 * ti[0], 16 * ti[1], 0, U, 16 * tj[0], 16 * tj[1], 0, lane, a0, a1);
        subtract(ti[0], tj[0], a0);
        subtract(ti[1], tj[1], a1);
    } else if (mine == 1) {
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
        acc = lds_mm16k<true, 64>(U, 16 * ti[0], 0, U, 16 * tj[0], 0, lane, acc);
        subtract(ti[0], tj[0], acc);
    }
}

__global__ void __launch_bounds__(256) chol_chain_kernel(const double* Kmat, double* L, double* Inv, long ld, long sBatch,
                                                         int nb, int* flags, long sFlags, int* info, int crow_mode,
                                                         int spin_limit, long long* trace, int merge_publish,
                                                         int kb = 0, int ke = -1, int late_polls = 3, int wt = 0,
                                                         int defer_publish = 0) {
    // wt: publish L_kk, inv_kk and L(k+1,k) as write-through stores, no L2 write-back per publication (wg_sync.hpp)
    // defer_publish (with wt): in a step whose next tiles were prefetched, leafdone[k] goes out behind the panel row's
    // products instead of in front of them -- the stores of L_kk / inv_kk drain under those products and the flag rides on
    // the barrier that follows them anyway (r05: the drain + barrier of the separate publication were 1.24 us per step on
    // the chain's critical path; the workers see inv_kk a few hundred nanoseconds later, in steps where the chain, not
    // they, is what the step waits for: their tiles were there two thirds into the leaf)
    // [kb, ke): the block columns this launch factors (two-level execution: one launch per super-panel; the tiles of
    // block kb then carry every earlier update by stream order, no flag).  Default: the whole matrix.
    if (ke < 0) ke = nb;
    // optional time stamps (100 MHz wall clock), 8 per step, for tools/chain_trace.py
#ifdef GPMPC_EMULATED
#define CHAIN_STAMP(i) ((void)0)
#else
#define CHAIN_STAMP(i) do { if (trace && threadIdx.x == 0) trace[((long)blockIdx.z * nb + k) * 8 + (i)] = wall_clock64(); } while (0)
#endif
    double* smem = GPMPC_DYN_SMEM();
    double* S = smem;
    double* T = S + 64 * LS;
    double* U = T + 64 * LS;
    double* Dr = U + 64 * LS;
    int* slot = (int*)(Dr + 64);
    double* P = Dr + 64 + 2;            // prefetched A(k+1,k), row stride LS
    double* Qp = P + 64 * LS;           // prefetched lower triangle of A(k+1,k+1), packed: (r, c) at r (r + 1) / 2 + c
    // The diagonal blocks of L and L^-1 go to memory as the 1056 column pairs that touch the lower triangle -- the rest
    // of those 64 x 64 blocks is zero from the workspace's allocation and nobody writes there (api_core.inl ws_alloc;
    // leaf64_kernel writes zeros) -- dealt evenly: 4-5 16-byte stores per thread and matrix instead of 8, a third less
    // traffic on the chain's critical path.  tri[e] = (r << 6) | 2c, built once.
    unsigned short* tri = (unsigned short*)(Qp + 2080);
    for (int e = threadIdx.x; e < CHAIN_TRI_PAIRS; e += 256) {
        int r = 0, m = e;
        while (m >= r / 2 + 1) { m -= r / 2 + 1; ++r; }
        tri[e] = (unsigned short)((r << 6) | (2 * m));
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long mb = (long)blockIdx.z * sBatch;
    const double* __restrict__ Kb = Kmat + mb;
    double* __restrict__ Lb = L + mb;
    double* __restrict__ Ib = Inv + mb;
    int* fl = flags + (long)blockIdx.z * sFlags;
    int* err = fl;
    int* leafdone = fl + 1;
    int* pan1 = fl + 1 + nb;
    int* tdone = fl + 1 + 2 * nb;

    for (int idx = tid; idx < 4096; idx += 256) {
        const int rr = idx >> 6, cc = idx & 63;
        S[rr * LS + cc] = (cc <= rr) ? Kb[(long)(64 * kb + rr) * ld + 64 * kb + cc] : 0.0;
        T[rr * LS + cc] = 0.0;
    }
    __syncthreads();
    int defer = 0;                                    // tiles of the current diagonal block's update left for the leaf's first panel
    for (int k = kb; k < ke; ++k) {
        const long o = (long)(64 * k) * ld + 64 * k;
        const long o10 = o + 64 * ld, o11 = o10 + 64;
        const unsigned tile_bytes = (unsigned)((63 * ld + 64) * 8);
        const wt_rsrc_t rL = wt_make_rsrc(Lb + o, tile_bytes), rI = wt_make_rsrc(Ib + o, tile_bytes),
                        rR = wt_make_rsrc(Lb + o10, tile_bytes);
        CHAIN_STAMP(0);
        // Prefetch for the second half of the step: if the two tiles A(k+1,k), A(k+1,k+1) carry the trailing update of
        // step k-1 two thirds into the leaf (they usually do), waves 2 and 3 -- idle from there on -- issue their loads and
        // put them into LDS (P, Qp) while wave 0 factors the last panel.  (The first version kept them in 64 registers of
        // every thread across the rest of the leaf; the compiler parked those in AGPRs, which means waiting for the loads
        // inside the leaf: 20.1 us per leaf against 17.9 without the prefetch.)
        bool pre = false;
        struct Prefetch {
            const double* Kb; long o10, o11, ld; const int* f0; const int* f1; int* slot; int tid, wave; bool active;
            bool* pre; double* P; double* Qp;
            double* S; const double* U; int defer, crow_mode;     // defer: 0 none, 1 = S holds A(k,k), 2 = Qp does
            int late_polls;
            long long* trace; int k;                 // (developer aid: the leaf's own time stamps, tools/chain_trace.py)
            __device__ __forceinline__ void stamp(int i) {
#ifndef GPMPC_EMULATED
                if (trace && tid == 0) trace[200000 + k * 16 + i] = wall_clock64();
#endif
            }
            // the six tiles of columns 16-63 of A(k,k) -= L(k,k-1) L(k,k-1)^T that the previous step left for now
            __device__ __forceinline__ void first() {
                if (defer) chain_diag_tiles(S, U, Qp, defer == 2, wave - 1, 3, 1, tid & 63, crow_mode);
            }
            // Called by every wave in front of the barrier behind the third panel -- which waves 2 and 3 reach right after
            // the barrier behind the second one, having nothing to do in between: the look at the hand-off flags is wave 3's
            // (thread 192), next to wave 0's third panel.  (r01-r03 had thread 0 do it: two dependent flag loads and the
            // acquire, 2-3 us, on the one wave every other wave of the leaf waits for.)  A few polls: the look comes a
            // panel earlier than it did.
            __device__ __forceinline__ void before() {
                if (active && tid == 192) {
                    int ok = !f0;
                    for (int it = 0; it < 3 && !ok; ++it) {
                        ok = flag_load(f0) >= 1 && flag_load(f1) >= 1;
                        if (!ok) __builtin_amdgcn_s_sleep(4);
                    }
                    if (ok && f0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    *slot = ok;
                }
            }
            __device__ __forceinline__ void after() {
                if (active && *slot != 0) *pre = true;
            }
            // waves 2 and 3, behind the barrier that precedes the last panel: 128 threads, 2048 double2 per tile, thread u
            // takes slots u + 128 i = (row 4 i + u / 32, columns 2 (u % 32) ..).  Loads and LDS stores sit in ONE region
            // without a barrier in between: nothing fetched is live across the leaf's synchronisation points.
            __device__ __forceinline__ void land() {
                // Second chance: the tiles often arrive a microsecond or two after the check in before() (hand-off latency
                // ~9 us against ~8 us from the publication to that check).  Waves 2 and 3 have nothing to do behind this
                // barrier anyway: each polls a little longer; what a wave decides is recorded in LDS (slot[2], slot[3])
                // and counts only if both agree -- the caller turns it into `pre` after the leaf.
                bool go = *pre;
                if (!go && active && f0) {
                    int ok = 0;
                    if ((tid & 63) == 0) {
                        for (int it = 0; it < late_polls && !ok; ++it) {
                            ok = flag_load(f0) >= 1 && flag_load(f1) >= 1;
                            if (!ok) __builtin_amdgcn_s_sleep(4);
                        }
                    }
                    ok = __builtin_amdgcn_readfirstlane(ok);
                    if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    if ((tid & 63) == 0) slot[wave] = ok;       // wave is 2 or 3
                    go = ok != 0;
                }
                if (go) {
                    const int u = tid - 128, r0 = u >> 5, c = (u & 31) * 2;
                    double2 pu[16], ps[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int rr = 4 * i + r0;
                        pu[i] = *reinterpret_cast<const double2*>(&Kb[o10 + (long)rr * ld + c]);
                        ps[i] = (c <= rr) ? *reinterpret_cast<const double2*>(&Kb[o11 + (long)rr * ld + c]) : double2{0.0, 0.0};
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int rr = 4 * i + r0;
                        P[rr * LS + c] = pu[i].x;
                        P[rr * LS + c + 1] = pu[i].y;
                        if (c <= rr) Qp[rr * (rr + 1) / 2 + c] = ps[i].x;
                        if (c + 1 <= rr) Qp[rr * (rr + 1) / 2 + c + 1] = ps[i].y;
                    }
                }
            }
        } pf{Kb, o10, o11, ld, k > kb ? &tdone[2 * (k - 1)] : nullptr, k > kb ? &tdone[2 * (k - 1) + 1] : nullptr, slot, tid, wave,
             k + 1 < ke, &pre, P, Qp, S, U, defer, crow_mode, late_polls, blockIdx.z == 0 ? trace : nullptr, k};
        defer = 0;
        if (tid == 0) { slot[2] = 0; slot[3] = 0; }       // (read after the leaf's barriers, written behind its third one)
        const int bad = leaf_body(S, T, U, Dr, 1, 15, crow_mode, pf);
        if (!pre && slot[2] != 0 && slot[3] != 0) pre = true;   // both loader waves saw the tiles late and landed them
        CHAIN_STAMP(1);
        if (tid == 0 && bad >= 0) atomicCAS(&info[blockIdx.z], 0, 64 * k + bad + 1);
        {   // two columns per thread and pair: 16-byte global stores.  r05: the (up to) five pairs of a thread in three
            // passes -- table look-ups, LDS reads, stores -- instead of five dependent look-up -> read -> store rounds (the
            // loop's 1.2 us were issue latency, not the drain: profiles/r05_chain_trace_leaf_restructured.txt)
            constexpr int NP = (CHAIN_TRI_PAIRS + 255) / 256;
            int rc[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) rc[q] = (tid + 256 * q < CHAIN_TRI_PAIRS) ? (int)tri[tid + 256 * q] : -1;
            double2 l[NP], v[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int rr = (rc[q] >> 6) & 63, cc = rc[q] & 63;    // cc <= rr; the second column may lie above the diagonal
                l[q].x = S[rr * LS + cc];
                l[q].y = (cc + 1 <= rr) ? S[rr * LS + cc + 1] : 0.0;
                v[q].x = T[rr * LS + cc];
                v[q].y = (cc + 1 <= rr) ? T[rr * LS + cc + 1] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if (rc[q] < 0) continue;
                const int rr = rc[q] >> 6, cc = rc[q] & 63;
                if (wt) {
                    wt_store16(rL, (unsigned)((rr * ld + cc) * 8), l[q]);
                    wt_store16(rI, (unsigned)((rr * ld + cc) * 8), v[q]);
                } else {
                    *reinterpret_cast<double2*>(&Lb[o + (long)rr * ld + cc]) = l[q];
                    *reinterpret_cast<double2*>(&Ib[o + (long)rr * ld + cc]) = v[q];
                }
            }
        }
        // merge_publish: leafdone[k] goes out together with pan1[k] a few microseconds later, saving one L2 write-back per
        // step on this critical path (off by default since r03: the courier and the workers' look-ahead want inv_kk early)
        // defer_leaf: leafdone[k] is then released by a bare flag_store behind the panel row's products further down.  That is
        // equivalent to wg_publish_wt (every wave drains its write-through stores, barrier, one lane stores the flag) ONLY IF
        // `pre` is workgroup-uniform (it is: read from LDS behind a barrier) and NO exit path lies between the stores above and
        // that flag_store -- the one early `return` of this loop, wg_wait2's time-out, sits in the !pre branch, which never
        // defers.  Whoever adds an exit path to the prefetched branch must publish leafdone[k] first.  (the soak
        // test and the bench line gate on handoff_timeouts == 0 with this on.)
        const bool defer_leaf = defer_publish && wt && !merge_publish && pre && k + 1 < ke;
        if (!defer_leaf && (!merge_publish || k + 1 == ke)) { if (wt) wg_publish_wt(&leafdone[k], 1); else wg_publish(&leafdone[k], 1); }
        CHAIN_STAMP(2);
        if (k + 1 == ke) break;
        const double* Asrc = P;                       // A(k+1,k): put there by the prefetch, else fetched now
        if (pre) {
            CHAIN_STAMP(3);
            // (the next diagonal block stays in Qp: the trailing update below reads it there and writes S -- no copy.
            //  The barrier that frees S for those writes is the one in front of the panel-row stores.)
        } else {
            // the two tiles below/right of the diagonal block must carry the trailing update of step k-1
            if (!wg_wait2(&tdone[2 * (k - 1)], 1, &tdone[2 * (k - 1) + 1], 1, err, spin_limit, slot, 1000000 + 1000 * k)) return;
            CHAIN_STAMP(3);
            double pu[16], ps[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int idx = tid + 256 * i, rr = idx >> 6, cc = idx & 63;
                pu[i] = Kb[o10 + (long)rr * ld + cc];
                ps[i] = (cc <= rr) ? Kb[o11 + (long)rr * ld + cc] : 0.0;
            }
            __syncthreads();                          // every thread is done reading S (the L_kk stores above)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int idx = tid + 256 * i, rr = idx >> 6, cc = idx & 63;
                P[rr * LS + cc] = pu[i];
                S[rr * LS + cc] = ps[i];
            }
        }
        __syncthreads();
        CHAIN_STAMP(4);
        // L_{k+1,k} = A_{k+1,k} inv_kk^T: wave w computes the 16-row strip w (4 tiles; inv_kk is lower triangular, so
        // tile tj needs depth 16 (tj + 1) only), straight from P into U
        // (r05: the four tiles share their A fragments and advance together -- one A read per step of 4 in K, the matrix
        //  instructions of the four independent accumulators back to back -- instead of four serial read-wait-multiply loops;
        //  every tile still sums its k in ascending order: same bits)
        d4 acc[4];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[tj] = d4{0.0, 0.0, 0.0, 0.0};
        {
            const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const double af = Asrc[(16 * wave + fr) * LS + 4 * q + fk];
#pragma unroll
                for (int tj = q / 4; tj < 4; ++tj) acc[tj] = mfma16(af, T[(16 * tj + fr) * LS + 4 * q + fk], acc[tj]);
            }
        }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) lds_put16(U, 16 * wave, 16 * tj, acc[tj], 1.0, lane, crow_mode);
        if (defer_leaf) GPMPC_DRAIN_VM();             // this wave's L_kk / inv_kk stores (issued before the products) have left
        __syncthreads();
        if (defer_leaf && tid == 0) flag_store(&leafdone[k], 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, rr = idx >> 5, cc = (idx & 31) * 2;
            double2 u;
            u.x = U[rr * LS + cc];
            u.y = U[rr * LS + cc + 1];
            if (wt) wt_store16(rR, (unsigned)((rr * ld + cc) * 8), u);
            else *reinterpret_cast<double2*>(&Lb[o10 + (long)rr * ld + cc]) = u;
        }
        CHAIN_STAMP(5);
        if (wt) { if (merge_publish) wg_publish_wt(&leafdone[k], 1, &pan1[k]); else wg_publish_wt(&pan1[k], 1); }
        else if (merge_publish) wg_publish(&leafdone[k], 1, &pan1[k]);
        else wg_publish(&pan1[k], 1);     // (contains the barrier that also orders the S loads above)
        CHAIN_STAMP(6);
        // A_{k+1,k+1} -= L_{k+1,k} L_{k+1,k}^T on the 10 lower 16 x 16 tiles: the four of the first 16 columns now, one
        // per wave -- the next leaf's first panel needs nothing else -- the other six by waves 1-3 while wave 0 factors
        // that panel (hook first(); 2.3 -> 0.6 us on the chain's critical path)
        chain_diag_tiles(S, U, Qp, pre, wave, 4, 0, lane, crow_mode);
        defer = pre ? 2 : 1;
        __syncthreads();
        CHAIN_STAMP(7);
    }
#undef CHAIN_STAMP
}

}  // namespace gpmpc
