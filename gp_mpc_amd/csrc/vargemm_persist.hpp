// Persistent products over a static tile schedule: the predictive variance and K^-1 = L^-T L^-1.
//
// PG_VAR   var_j = sf2 - sum_i (L^-1 ks_j)_i^2 for every test point j  (a9, gp_functions.py:118-126; GP.covar gp_class.py:377-380):
//          A = L^-1 (lower triangular, K contiguous), B = KsT (K contiguous), column sums of squares per 128-row tile.  Same
//          arithmetic, tiles and LDS images as gemm_f64_dma_kernel<128,128,2,4,2,4>: the per-tile sums are bit-identical to
//          that kernel's.  Fused into the same epilogue (GemmP::wvec / partm): the mean ks_j^T alpha = (L^-1 ks_j)^T (L^-1 y)
//          as partial sums sum_i V_ij w_i -- no second pass over the 328 MB of cross-covariances, which as a kernel of its
//          own next to this one cost the C2 step 70 us.
// PG_XTX   lower triangle of K^-1 = X^T X, X = L^-1  (a6, optimize.py:489-490): A = B = XT, the transposed copy of L^-1 made by
//          transpose_lower_kernel, so that both operands are K contiguous (the M/N-contiguous instantiation of the one-tile
//          kernel reads its fragments with twice the LDS instructions and spills); tile (tm >= tn) sums over k >= 128 tm.
//
// The 128 x 128 tiles are not handed out by the hardware dispatcher:
//   * The variance tile in block row tm is tm + 1 K-units long, so the 2528 tiles of C2 (32 x 79) carry 1 ... 32 units and
//     the 512 workgroup slots of the chip get ~5 tiles each.  Workgroups are dealt to the 8 XCDs round-robin whatever
//     their load, and 79 columns on 8 XCDs leave one XCD with nine tenths of the others' work: in-order dispatch ends
//     4.3 % above the mean slot load (simulation of the dispatcher, heavy rows first), i.e. the last 0.1 ms of the
//     kernel run on a draining chip.  Here 2 x CUs workgroups stay resident and walk lists made on the host
//     (persist_schedule: the XCDs are levelled first by moving a few tiles -- the only ones that leave the XCD that holds the
//     rest of their Ks panel --, then inside every XCD longest tile first to the least loaded slot and moves / swaps off the
//     heaviest slot): within 0.5 % of the mean.  (A first version balanced across XCDs freely: 16 % of the tiles ran away
//     from their panel and the kernel fetched 43 % more than the dispatcher's order, profiles/r04_*.)  The doubly triangular
//     K^-1 product is worse off with the dispatcher (row tm holds tm + 1 tiles of T - tm units and as many workgroups that exit at once).
//   * The slabs of a workgroup's tiles form one stream through the two-image ring: the first slab of the next tile is
//     requested behind the barrier of the current tile's last step, so a tile boundary costs the epilogue and nothing
//     else (the one-tile kernel drains its ring, writes its sums, exits, and its successor starts with an empty ring).
//     The epilogue's scratch has its own LDS for that reason (6 KB: two sets of partial sums, w of two tiles), and its
//     barrier waits for LDS traffic only.
#pragma once
#include <algorithm>
#include <vector>
#include "gemm_f64_dma.hpp"

namespace gpmpc {

// tile word: batch index z (8 bits) | block row tm (12 bits) | block column tn (12 bits)
constexpr int VAR_TILE = 128;
// amdgpu_num_vgpr counts halves of the unified 512-entry file of gfx90a+ (the backend doubles the request): 60 = a budget of
// 120 registers (what does not fit is reloaded at tile switches only), so that four waves per SIMD leave 32 per lane and the
// alpha kernels of the workers' queue, which run NEXT TO the variance product, still find room on its CUs.  r02-r05 asked for
// 58 = 116: registers are allocated in granules of 8, so that occupied 120 as well and only cost spills (8 VGPRs, 36 B of
// scratch); 61 / 62 occupy 128 and lock the alpha kernels out until the product ends (profiles/r06_var_vgprs_ab.txt:
// 58 / 59 / 60 / 61 / 62 -> step 4.054 / 4.102 / 4.037 / 4.041 / 4.056 ms, alpha 0.09 / 0.08 / 0.08 / 0.91 / 1.19 ms).
constexpr int VAR_VGPRS = 60;
__host__ __device__ inline int var_tile_word(int z, int tm, int tn) { return (z << 24) | (tm << 12) | tn; }

enum { PG_VAR = 0, PG_XTX = 1 };

template <int MODE>
__device__ __forceinline__ void persist_gemm_body(const GemmP& p, const int* __restrict__ list, const int* __restrict__ off,
                                                  const int* __restrict__ zmap) {
    constexpr int BM = VAR_TILE, BN = VAR_TILE, BK = 16, WGM = 2, WGN = 4, NW = WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
    constexpr int LA = BM / 8, LB = BN / 8, LPW = (LA + LB) / NW;
    constexpr int IMG_A = BM * 128, SLAB = (BM + BN) * 128;
    char* smem = (char*)GPMPC_DYN_SMEM();
    double* red = reinterpret_cast<double*>(smem + 2 * SLAB);      // [WGM][BN], not part of the ring
    double* red2 = red + WGM * BN;                                 // ... the same for the mean's partial sums
    double* wl = red2 + WGM * BN;                                  // [2][BM]: w of the current tile's rows (two tiles in turn)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int beg = off[blockIdx.x], end = off[blockIdx.x + 1];
    if (beg >= end) return;

    const int fr = lane & 15, fq = lane >> 4;
    const unsigned sw = (unsigned)((fr >> 1) & 7);
    unsigned fa[2], fb[2];                                         // fragment offsets inside an image (gemm_f64_dma.hpp)
    fa[0] = (unsigned)((wm * WM + fr) * 128) + (((unsigned)fq ^ sw) << 4);
    fa[1] = fa[0] ^ 64u;
    fb[0] = (unsigned)(IMG_A + (wn * WN + fr) * 128) + (((unsigned)fq ^ sw) << 4);
    fb[1] = fb[0] ^ 64u;

    // ---- request cursor: tile ri of the list, its next slab rt of rnk
    int ri = beg, rt = 0, rnk = 0, rk0 = 0;
    unsigned vo[LPW];
    dma_rsrc_t rsA, rsB;
    auto open_request = [&](int word) {
        const int zw = (int)((unsigned)word >> 24), m0 = ((word >> 12) & 4095) * BM, n0 = (word & 4095) * BN;
        const int z = (MODE == PG_XTX && zmap) ? zmap[zw] : zw;     // (zmap: the matrices of a subset of a batch)
        if (MODE == PG_VAR) { rk0 = 0; rnk = (min(p.K, m0 + BM) + BK - 1) / BK; }
        else { rk0 = m0; rnk = (p.K - m0) / BK; }                  // (K is a multiple of 16: gemm_dma_supported)
        rsA = dma_make_rsrc(p.A + (long)z * p.sA, (unsigned)((long)p.M * p.lda * 8));
        rsB = dma_make_rsrc(p.B + (long)z * p.sB, (unsigned)((long)p.N * p.ldb * 8));
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int j = wave + NW * i;                           // [0, LA): piece j of the A image, else of the B image
            const bool isA = j < LA;
            const int row = 8 * (isA ? j : j - LA) + (lane >> 3);
            const unsigned piece = (unsigned)(lane & 7) ^ (unsigned)((row >> 1) & 7);
            const long grow = isA ? (long)min(m0 + row, p.M - 1) * p.lda : (long)min(n0 + row, p.N - 1) * p.ldb;
            vo[i] = (unsigned)(grow * 8) + (piece << 4);
        }
    };
    // (the pointer form of dma_load16 here: with the integer form and its M0 clobber this kernel measured 0.7 % slower,
    //  profiles/r05_dma_lds_address_ab.txt -- it runs inside a 116-register budget and at 96 % matrix-pipe duty)
    auto request_next = [&](int g) {                               // next slab of the stream -> image g & 1
        if (ri >= end) return;
        char* img = smem + (g & 1) * SLAB;
        const unsigned k0 = (unsigned)(rk0 + rt * BK) * 8u;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int j = wave + NW * i;
            if (j < LA) dma_load16(rsA, img + 1024 * j, vo[i], k0);
            else dma_load16(rsB, img + 1024 * j, vo[i], k0);
        }
        if (++rt == rnk) {
            rt = 0;
            if (++ri < end) open_request(list[ri]);
        }
    };

    // ---- compute cursor
    // a wave whose 64 rows lie entirely above (PG_VAR: A(m,k) = 0 for k > m) or below (PG_XTX: A(m,k) = 0 for k < m) a slab's
    // K range multiplies exact zeros there and sits the slab out: it works on the slabs [kslo, kshi) of its tile
    int ci = beg, ct = 0, cword = list[beg], cnk, kslo, kshi;
    auto open_compute = [&](int word, int par) {
        const int m0 = ((word >> 12) & 4095) * BM;
        if (MODE == PG_VAR && p.partm) {
            // w of the tile's rows for the fused mean: read in the epilogue, at least eight ring barriers from here; the buffer
            // of the previous tile may still be read by a slower wave's epilogue, hence two of them
            if (tid < BM) wl[par * BM + tid] = (m0 + tid < p.M) ? p.wvec[(long)((unsigned)word >> 24) * p.sWv + m0 + tid] : 0.0;
            lds_flush();
        }
        if (MODE == PG_VAR) {
            cnk = (min(p.K, m0 + BM) + BK - 1) / BK;
            kslo = 0;
            kshi = (m0 + (wm + 1) * WM + BK - 1) / BK;
        } else {
            cnk = (p.K - m0) / BK;
            kslo = wm * WM / BK;
            kshi = 1 << 30;
        }
    };
    open_compute(cword, ci & 1);

    d4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    open_request(cword);
    request_next(0);
    for (int g = 0;; ++g) {
        dma_wait<0>();                                             // slab g has landed (this wave's pieces)
        dma_barrier();                                             // ... everybody's, and image (g + 1) & 1 is free
        request_next(g + 1);
        if (ct >= kslo && ct < kshi) {
            const char* img = smem + (g & 1) * SLAB;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double2 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const double2*>(img + fa[h] + i * 2048);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const double2*>(img + fb[h] + j * 2048);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(a[i].x, b[j].x, acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(a[i].y, b[j].y, acc[i][j]);
            }
        }
        if (++ct < cnk) continue;

        // ---- tile finished: its epilogue, then the next tile of the list
        const int zw = (int)((unsigned)cword >> 24), tm = (cword >> 12) & 4095, m0 = tm * BM, n0 = (cword & 4095) * BN;
        const int z = (MODE == PG_XTX && zmap) ? zmap[zw] : zw;
        if (MODE == PG_VAR) {                                      // column sums of squares over the tile's rows < M
            const bool with_mean = p.partm != nullptr;             // ... and the mean's partial sums sum_m V[m][n] w[m]
            const double* wt = wl + (ci & 1) * BM + wm * WM;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                double s = 0.0, s2 = 0.0;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ml = i * 16 + crow(lane, r, p.crow_mode);
                        const double v = (m0 + wm * WM + ml < p.M) ? acc[i][j][r] : 0.0;
                        s += v * v;
                        if (with_mean) s2 = fma(v, wt[ml], s2);
                        acc[i][j][r] = 0.0;
                    }
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                if (lane < 16) red[wm * BN + wn * WN + j * 16 + lane] = s;
                if (with_mean) {
                    s2 += __shfl_xor(s2, 16);
                    s2 += __shfl_xor(s2, 32);
                    if (lane < 16) red2[wm * BN + wn * WN + j * 16 + lane] = s2;
                }
            }
            lds_barrier();                                         // (the slab in flight is not waited for)
            if (tid < BN && n0 + tid < p.N) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < WGM; ++w) t += red[w * BN + tid];
                p.part[(long)z * p.sPart + (long)tm * p.ldpart + n0 + tid] = t;
                if (with_mean) {
                    double t2 = 0.0;
#pragma unroll
                    for (int w = 0; w < WGM; ++w) t2 += red2[w * BN + tid];
                    p.partm[(long)z * p.sPart + (long)tm * p.ldpart + n0 + tid] = t2;
                }
            }
            // (`red` is written again after >= 8 more ring barriers: no barrier needed behind its readers)
        } else {                                                   // the lower triangle of the product
            double* __restrict__ C = p.C + (long)z * p.sC;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + wm * WM + i * 16 + crow(lane, r, p.crow_mode);
                        const int n = n0 + wn * WN + j * 16 + fr;
                        if (m < p.M && n <= m) C[(long)m * p.ldc + n] = acc[i][j][r];
                        acc[i][j][r] = 0.0;
                    }
        }
        if (++ci >= end) break;
        cword = list[ci];
        ct = 0;
        open_compute(cword, ci & 1);
    }
}

__global__ void __launch_bounds__(512, 4) __attribute__((amdgpu_num_vgpr(VAR_VGPRS))) vargemm_persist_kernel(GemmP p, const int* __restrict__ list,
                                                                                                            const int* __restrict__ off) {
    persist_gemm_body<PG_VAR>(p, list, off, nullptr);
}
__global__ void __launch_bounds__(512, 4) xtx_persist_kernel(GemmP p, const int* __restrict__ list, const int* __restrict__ off,
                                                             const int* __restrict__ zmap) {
    persist_gemm_body<PG_XTX>(p, list, off, zmap);
}

// ---- host side: the schedule ------------------------------------------------------------------------
// Slots are the workgroups of the launch; slot s runs on XCD s % nx.  Cost of a tile in half slabs: 2 nk + 1 (the
// epilogue).  Returns the slot lists concatenated (tile words) and their offsets [slots + 1].
struct PersistTile { int cost, word, x; };                        // x: home XCD (the one that holds the rest of its operand panel)
struct VarSchedule {
    std::vector<int> list, off;
    double mean_load = 0.0, max_load = 0.0;
    int home = 0;                                                  // tiles on their home XCD
};

inline VarSchedule persist_schedule(std::vector<PersistTile>& tiles, int slots, int nx = 8) {
    typedef PersistTile T;
    if (nx > slots || slots % nx) nx = 1;
    const int per = slots / nx;                                    // slots of an XCD: x, x + nx, ...
    std::stable_sort(tiles.begin(), tiles.end(), [](const T& a, const T& b) { return a.cost > b.cost; });
    long total = 0;
    std::vector<long> tot(nx, 0);
    std::vector<std::vector<T>> byx(nx);
    for (const T& t : tiles) { total += t.cost; tot[t.x % nx] += t.cost; byx[t.x % nx].push_back(t); }
    // 1. level the XCDs: the largest tile that fits the gap goes from the fullest to the emptiest XCD.  Only these tiles leave
    //    the XCD that holds the rest of their operand panel (C2: 79 columns on 8 XCDs -> about twenty of 2528).
    const double xtarget = (double)total / nx;
    for (int it = 0; it < (int)tiles.size(); ++it) {
        const int xo = (int)(std::max_element(tot.begin(), tot.end()) - tot.begin());
        const int xu = (int)(std::min_element(tot.begin(), tot.end()) - tot.begin());
        const double gap = std::min((double)tot[xo] - xtarget, xtarget - (double)tot[xu]);
        int pick = -1;
        for (int i = 0; i < (int)byx[xo].size(); ++i)
            if ((double)byx[xo][i].cost <= gap) { pick = i; break; }   // (sorted: the first that fits is the largest)
        if (pick < 0) break;
        const T t = byx[xo][pick];
        byx[xo].erase(byx[xo].begin() + pick);
        auto pos = std::lower_bound(byx[xu].begin(), byx[xu].end(), t, [](const T& a, const T& b) { return a.cost > b.cost; });
        byx[xu].insert(pos, t);
        tot[xo] -= t.cost;
        tot[xu] += t.cost;
    }
    // 2. inside every XCD: longest tile first to its least loaded slot, then moves / swaps off its heaviest slot
    std::vector<long> load(slots, 0);
    std::vector<std::vector<T>> lists(slots);
    for (int x = 0; x < nx; ++x) {
        for (const T& t : byx[x]) {
            int s = x;
            for (int c = x; c < slots; c += nx)
                if (load[c] < load[s]) s = c;
            load[s] += t.cost;
            lists[s].push_back(t);
        }
        const double target = (double)tot[x] / per;
        for (int it = 0; it < 8 * per; ++it) {
            int smax = x;
            for (int c = x; c < slots; c += nx)
                if (load[c] > load[smax]) smax = c;
            if ((double)load[smax] <= 1.002 * target) break;
            long best = 0;
            int bi = -1, bs = -1, bj = -1;
            for (int s2 = x; s2 < slots; s2 += nx) {
                if (s2 == smax || load[s2] >= load[smax]) continue;
                for (int i = 0; i < (int)lists[smax].size(); ++i) {
                    const long c = lists[smax][i].cost;
                    if (load[s2] + c < load[smax]) {               // move
                        const long gain = load[smax] - std::max(load[smax] - c, load[s2] + c);
                        if (gain > best) { best = gain; bi = i; bs = s2; bj = -1; }
                    }
                    for (int j = 0; j < (int)lists[s2].size(); ++j) {   // swap
                        const long dlt = c - lists[s2][j].cost;
                        if (dlt > 0 && load[s2] + dlt < load[smax]) {
                            const long gain = load[smax] - std::max(load[smax] - dlt, load[s2] + dlt);
                            if (gain > best) { best = gain; bi = i; bs = s2; bj = j; }
                        }
                    }
                }
            }
            if (bi < 0) break;
            if (bj < 0) {
                const T t = lists[smax][bi];
                lists[smax].erase(lists[smax].begin() + bi);
                lists[bs].push_back(t);
                load[smax] -= t.cost;
                load[bs] += t.cost;
            } else {
                std::swap(lists[smax][bi], lists[bs][bj]);
                const long dlt = lists[bs][bj].cost - lists[smax][bi].cost;
                load[smax] -= dlt;
                load[bs] += dlt;
            }
        }
    }
    VarSchedule r;
    r.off.resize(slots + 1);
    r.mean_load = (double)total / slots;
    for (int s = 0; s < slots; ++s) {
        std::stable_sort(lists[s].begin(), lists[s].end(), [](const T& a, const T& b) { return a.cost > b.cost; });
        r.off[s] = (int)r.list.size();
        for (const T& t : lists[s]) {
            r.list.push_back(t.word);
            if (t.x % nx == s % nx) ++r.home;
        }
        r.max_load = std::max(r.max_load, (double)load[s]);
    }
    r.off[slots] = (int)r.list.size();
    return r;
}

// variance product: tilesM x tilesN tiles per matrix, the tile in block row tm is min(K, 128 (tm + 1)) / 16 slabs long;
// home XCD by column (the Ks panel)
inline VarSchedule var_schedule(int tilesM, int tilesN, int batch, int K, int slots, int nx = 8) {
    std::vector<PersistTile> tiles;
    tiles.reserve((size_t)tilesM * tilesN * batch);
    for (int z = 0; z < batch; ++z)
        for (int tm = 0; tm < tilesM; ++tm) {
            const int nk = (std::min(K, (tm + 1) * VAR_TILE) + 15) / 16;
            for (int tn = 0; tn < tilesN; ++tn) tiles.push_back(PersistTile{2 * nk + 1, var_tile_word(z, tm, tn), (tn + z) % nx});
        }
    return persist_schedule(tiles, slots, nx);
}

// K^-1 = X^T X, lower triangle: tiles (tm >= tn), each (K - 128 tm) / 16 slabs long; home XCD by block row (the tm + 1 tiles
// of a row are equally long and share their A panel)
inline VarSchedule xtx_schedule(int tilesM, int batch, int K, int slots, int nx = 8) {
    std::vector<PersistTile> tiles;
    tiles.reserve((size_t)tilesM * (tilesM + 1) / 2 * batch);
    for (int z = 0; z < batch; ++z)
        for (int tm = 0; tm < tilesM; ++tm) {
            const int nk = (K - tm * VAR_TILE) / 16;
            for (int tn = 0; tn <= tm; ++tn) tiles.push_back(PersistTile{2 * nk + 1, var_tile_word(z, tm, tn), (tm + z) % nx});
        }
    return persist_schedule(tiles, slots, nx);
}

// the device copy of a schedule, cached on the model handle
struct VarSchedDev {
    int mode = 0, tilesM = 0, tilesN = 0, batch = 0, K = 0, slots = 0;
    int* list = nullptr;
    int* off = nullptr;
};

template <int MODE>
inline void launch_persist_gemm(const GemmP& p, const VarSchedDev& s, hipStream_t stream, const int* zmap = nullptr) {
    constexpr int lds = 2 * (VAR_TILE + VAR_TILE) * 128 + (2 + 2 + 2) * VAR_TILE * 8 + 16;   // ring, red, red2, wl
    static bool attr_set = false;
    if (!attr_set) {
        const void* fn = MODE == PG_VAR ? reinterpret_cast<const void*>(&vargemm_persist_kernel) : reinterpret_cast<const void*>(&xtx_persist_kernel);
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    if (MODE == PG_VAR) hipLaunchKernelGGL(vargemm_persist_kernel, dim3(s.slots), dim3(512), lds, stream, p, (const int*)s.list, (const int*)s.off);
    else hipLaunchKernelGGL(xtx_persist_kernel, dim3(s.slots), dim3(512), lds, stream, p, (const int*)s.list, (const int*)s.off, zmap);
}

}  // namespace gpmpc
