// The small levels of a row panel's triangular inverse, following the chain block by block in ONE resident launch.
//
// The level-batched trtri_range (api_factor.inl) starts when the whole panel is factored and is then two dependent launches
// per level; its first three levels (nodes of 128, 256 and 512 rows) are six latency-bound launches of 6-40 us each: 56 us
// for the first panel, 140-160 us for the others next to the products of the inverse -- in front of every product that needs
// I_i, and behind the chain for the last panel (r04 timeline).  A launch per node that waits on the chain's flags is no way
// out: a flag-gated launch costs 7-40 us on its queue and a panel has thirty of them (measured).
// Here a handful of workgroups stay resident next to the worker launch that factors the panel and walk a list of 64 x 64 tile
// products made on the host (trtri_follow_tasks), every node [L11 0; L21 L22] of the tree up to `smax` rows per child as
//     W    = L21 inv11        tile (ti, tj) = sum_{k >= tj} L21(ti, k) inv11(k, tj)       as soon as the left child is inverted and
//                                                                                         its last block column is final
//     inv21 = -inv22 W        tile (ti, tj) = -sum_{k <= ti} inv22(ti, k) W(k, tj)        as soon as the right child is inverted
// with the dependencies as flags / counters in global memory: the chain's leafdone / pan1 and the workers' colready for what
// the factorisation delivers, one counter per node and phase for what the tasks deliver to each other (agent-scope release /
// acquire, wg_sync.hpp).  Task t belongs to workgroup t mod G and every workgroup takes its tasks in list order, which is a
// topological order: no cycle of waits can form as long as all G workgroups are resident (they are launched behind the gate
// of their worker launch, a few of them, with 16 KB of LDS so that they do not land on the chain's or a worker's CU).
// When the panel's last leaf is out one inv21 per level is left, a few microseconds each; trtri_range takes over at 2 smax rows.
// Operands come straight from L2 / memory into the matrix instruction's fragment layout (no LDS staging: the tiles were just
// written by another CU, and a product is 64 x 64 x 64 k).
#pragma once
#include <algorithm>
#include <vector>
#include "chol_chain.hpp"
#include "mfma_f64.hpp"
#include "wg_sync.hpp"

namespace gpmpc {

constexpr int TRF_INTS = 24;          // ints per task
constexpr int TRF_LDS_BYTES = 16384;  // (placement only: see above)
// task words
enum { TRF_KIND = 0, TRF_AROW, TRF_ACOL, TRF_BROW, TRF_BCOL, TRF_KLO, TRF_KHI, TRF_CROW, TRF_CCOL, TRF_WOFF, TRF_S,
       TRF_F0, TRF_F1, TRF_F2, TRF_C0, TRF_C0T, TRF_C1, TRF_C1T, TRF_DONE };

__host__ __device__ inline int chain_trtri_index(int nb) { return 1 + 8 * nb + 520; }   // 2 nb node counters behind the handed[] words

// kind 0: C = W slot (ld s) <- A = L (ld), B = Inv (ld);   kind 1: C = Inv (ld) <- -A = Inv (ld), B = W slot (ld s)
__global__ void __launch_bounds__(256) trtri_follow_kernel(const double* __restrict__ L, double* __restrict__ Inv, double* __restrict__ W,
                                                           long ld, const int* __restrict__ tasks, int ntasks, int* flags, int crow_mode,
                                                           int spin_limit) {
    int* slot = reinterpret_cast<int*>(GPMPC_DYN_SMEM());
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fk = lane >> 4;
    for (int t = blockIdx.x; t < ntasks; t += gridDim.x) {
        const int* tk = tasks + (long)t * TRF_INTS;
        // ---- wait: up to three flags (>= 1) and two counters (>= target); one lane polls, bounded
        if (tid == 0) {
            int ok = 1, spins = 0;
            for (;;) {
                bool ready = true;
                for (int q = 0; q < 3 && ready; ++q) {
                    const int f = tk[TRF_F0 + q];
                    if (f >= 0 && flag_load(flags + f) < 1) ready = false;
                }
                if (ready && tk[TRF_C0] >= 0 && flag_load(flags + tk[TRF_C0]) < tk[TRF_C0T]) ready = false;
                if (ready && tk[TRF_C1] >= 0 && flag_load(flags + tk[TRF_C1]) < tk[TRF_C1T]) ready = false;
                if (ready) break;
                __builtin_amdgcn_s_sleep(4);
                if (flag_load(flags) != 0) { ok = 0; break; }
                if (++spins > spin_limit) { flag_store(flags, 7000000 + t); ok = 0; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            *slot = ok;
        }
        __syncthreads();
        const bool go = *slot != 0;
        __syncthreads();
        if (!go) return;
        const int kind = tk[TRF_KIND], s = tk[TRF_S];
        const double* __restrict__ A = kind == 0 ? L : Inv;
        const double* __restrict__ B = kind == 0 ? Inv : W + tk[TRF_WOFF];
        const long ldb = kind == 0 ? ld : (long)s;
        const long arow = tk[TRF_AROW], acol = tk[TRF_ACOL], brow = tk[TRF_BROW], bcol = tk[TRF_BCOL];
        d4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
        for (int k = tk[TRF_KLO]; k <= tk[TRF_KHI]; ++k) {
            const double* __restrict__ Ak = A + (arow + 32 * wr + fr) * ld + acol + 64 * k + fk;
            const double* __restrict__ Bk = B + (brow + 64 * k + fk) * ldb + bcol + 32 * wc + fr;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                double a[8][2], b[8][2];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int kk = 8 * half + q;
                    a[q][0] = Ak[4 * kk];
                    a[q][1] = Ak[16 * ld + 4 * kk];
                    b[q][0] = Bk[(long)(4 * kk) * ldb];
                    b[q][1] = Bk[(long)(4 * kk) * ldb + 16];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a[q][i], b[q][j], acc[i][j]);
            }
        }
        double* __restrict__ C = kind == 0 ? W + tk[TRF_WOFF] : Inv;
        const long ldc = kind == 0 ? (long)s : ld;
        const double sign = kind == 0 ? 1.0 : -1.0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    C[(tk[TRF_CROW] + 32 * wr + 16 * i + crow(lane, r, crow_mode)) * ldc + tk[TRF_CCOL] + 32 * wc + 16 * j + fr] = sign * acc[i][j][r];
        // ---- publish: everybody's stores drained, then one release + counter bump
        GPMPC_DRAIN_VM();
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            GPMPC_DRAIN_VM();
            __hip_atomic_fetch_add(flags + tk[TRF_DONE], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- host side: the task list of a panel ----------------------------------------------------------------
// Rows [base0, base0 + n) of the matrix (multiples of 64), nodes with a left child of up to smax rows; nb = Np / 64.  W slots from
// wbase (doubles into the scratch), *wneed = what they take.  The list is ordered by the block whose completion enables a task
// (inv21 before the W it enables, small nodes first); `counters` = node counters used, two per node from chain_trtri_index(nb).
struct TrfList {
    std::vector<int> tasks;
    long wneed = 0;
    int ntasks = 0, counters = 0;
};

inline TrfList trtri_follow_tasks(int nb, long base0, int n, int smax, long wbase, int counter0) {
    struct Node { long base; int s, h2, id; long wo; };
    std::vector<Node> nodes;
    long wo = wbase;
    int nid = 0;
    for (int s = 64; s < n && s <= smax; s *= 2)
        for (long b = 0; b + s < n; b += 2 * (long)s) {
            const int h2 = (int)std::min<long>(s, n - b - s);
            nodes.push_back(Node{base0 + b, s, h2, nid++, wo});
            wo += (long)h2 * s;
        }
    auto find_node = [&](long base, int rows) -> const Node* {       // the node that spans exactly [base, base + rows)
        for (const Node& nd : nodes)
            if (nd.base == base && nd.s + nd.h2 == rows) return &nd;
        return nullptr;
    };
    const int cbase = chain_trtri_index(nb) + 2 * counter0;
    struct Op { int trig, kind, s; std::vector<int> words; };
    std::vector<Op> ops;
    for (const Node& nd : nodes) {
        const int st = nd.s / 64, ht = nd.h2 / 64;
        const int c_w = cbase + 2 * nd.id, c_inv = cbase + 2 * nd.id + 1;
        // completion of the children: a single block -> its leaf; a node -> its inv21 counter
        const Node* lch = nd.s > 64 ? find_node(nd.base, nd.s) : nullptr;
        const Node* rch = nd.h2 > 64 ? find_node(nd.base + nd.s, nd.h2) : nullptr;
        const int trig_w = (int)((nd.base + nd.s) / 64 - 1), trig_i = (int)((nd.base + nd.s + nd.h2) / 64 - 1);
        for (int ti = 0; ti < ht; ++ti)
            for (int tj = 0; tj < st; ++tj) {
                std::vector<int> w(TRF_INTS, -1);                  // W tile
                w[TRF_KIND] = 0; w[TRF_S] = nd.s; w[TRF_WOFF] = (int)nd.wo;
                w[TRF_AROW] = (int)(nd.base + nd.s) + 64 * ti; w[TRF_ACOL] = (int)nd.base;
                w[TRF_BROW] = (int)nd.base; w[TRF_BCOL] = (int)nd.base + 64 * tj;
                w[TRF_KLO] = tj; w[TRF_KHI] = st - 1;
                w[TRF_CROW] = 64 * ti; w[TRF_CCOL] = 64 * tj;
                w[TRF_F0] = chain_pan1_index(nb, trig_w);
                w[TRF_F1] = nb - trig_w - 2 > 0 ? chain_colready_index(nb, trig_w) : -1;
                if (lch) { w[TRF_C0] = cbase + 2 * lch->id + 1; w[TRF_C0T] = (lch->s / 64) * (lch->h2 / 64); }
                else w[TRF_F2] = 1 + trig_w;                       // leafdone of the single left block (implied by pan1, kept explicit)
                w[TRF_DONE] = c_w;
                ops.push_back(Op{trig_w, 1, nd.s, w});
                std::vector<int> v(TRF_INTS, -1);                  // inv21 tile
                v[TRF_KIND] = 1; v[TRF_S] = nd.s; v[TRF_WOFF] = (int)nd.wo;
                v[TRF_AROW] = (int)(nd.base + nd.s) + 64 * ti; v[TRF_ACOL] = (int)(nd.base + nd.s);
                v[TRF_BROW] = 0; v[TRF_BCOL] = 64 * tj;
                v[TRF_KLO] = 0; v[TRF_KHI] = ti;
                v[TRF_CROW] = (int)(nd.base + nd.s) + 64 * ti; v[TRF_CCOL] = (int)nd.base + 64 * tj;
                v[TRF_C0] = c_w; v[TRF_C0T] = st * ht;
                if (rch) { v[TRF_C1] = cbase + 2 * rch->id + 1; v[TRF_C1T] = (rch->s / 64) * (rch->h2 / 64); }
                v[TRF_F0] = 1 + trig_i;                            // leafdone of the right child's last block
                v[TRF_DONE] = c_inv;
                ops.push_back(Op{trig_i, 0, nd.s, v});
            }
    }
    std::stable_sort(ops.begin(), ops.end(), [](const Op& x, const Op& y) {
        if (x.trig != y.trig) return x.trig < y.trig;
        if (x.kind != y.kind) return x.kind < y.kind;
        return x.s < y.s;
    });
    TrfList r;
    r.wneed = wo - wbase;
    r.ntasks = (int)ops.size();
    r.counters = nid;
    r.tasks.reserve(ops.size() * TRF_INTS);
    for (const Op& o : ops) r.tasks.insert(r.tasks.end(), o.words.begin(), o.words.end());
    return r;
}

}  // namespace gpmpc
