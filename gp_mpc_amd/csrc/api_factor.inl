// api_factor.inl -- part of gpmpc_api.hip (one translation unit; included in order, not compiled alone).
// Concern: factor scheduling: triangular inverse (level-batched / pipelined), the executions of the blocked Cholesky (single queue, chain + flagged GEMMs, chain + tile-owner workers, two-level panels), alpha, K^-1.

static GemmP gemm_base(const Ctx& cx) {
    GemmP p;
    std::memset(&p, 0, sizeof(p));
    p.alpha = 1.0;
    p.crow_mode = cx.crow_mode;
    return p;
}

static int g_vargemm_persist = -1;  // gpmpc_set_tuning("vargemm_persist", 0 / 1 / 2): dispatcher order / static schedule / ... at any size; -1: GPMPC_VARGEMM_PERSIST or default
static int g_worker_courier = -1;   // gpmpc_set_tuning("worker_courier", 0 / 1): tile owners only / with the courier; -1: GPMPC_COURIER or default

// Fit factorisation = right-looking blocked Cholesky (NB = 64) + level-by-level batched triangular
// inverse.  K is consumed (trailing updates in place), L and Inv = L^-1 are written; [batch][Np x Np].
//
//   for each 64-column panel k:   leaf: L_kk = chol(A_kk), inv_kk = L_kk^-1        (one workgroup)
//                                 panel: L21 = A21 inv_kk^T                         (MFMA)
//                                 trailing: A22 -= L21 L21^T  (lower)               (MFMA)
//   then for s = 64, 128, ...:    every node [L11 0; L21 L22] with |L11| = s in ONE batched launch pair:
//                                 W = L21 inv11,  inv21 = -inv22 W                  (MFMA GEMMs)
//
// Three executions of this algorithm, chosen per call by factor_chain / its caller:
//   * factor_blocked: one queue, three launches per panel (fallback, gpmpc_cholesky, gpmpc_append);
//   * factor_chain with flagged GEMM launches: the leaf / row k+1 / diagonal-tile chain in ONE persistent
//     workgroup (chol_chain.hpp), panel and trailing GEMMs on a side queue coupled through flags, the
//     inverse pipelined behind the chain on a third queue (trtri_segment);
//   * factor_chain with tile-owner workers (chol_worker.hpp): the trailing matrix lives in the registers
//     of persistent workgroups, in two launches so that the CUs the second one leaves free invert the left
//     half while the chain finishes.
// (The first version recursed on [L11 0; L21 L22] with the inverse products inside the recursion: 4
// latency-bound launches per node on the chain, 5.5 ms at N = 4096; a plain second stream next to the
// single-queue version did not pay because the leaf slows down 3-8x when it shares a CU with MFMA waves.)

// level-by-level batched inverse of the diagonal range [base, base + n) (rows), given its 64-blocks
// (scratch / sScratch: the W = L21 inv11 products of a level go there instead of to the head of ws.W -- a caller whose
//  other queues use ws.W at the same time, factor_twolevel)
static void trtri_range(const Ctx& cx, Workspace& ws, hipStream_t stream, long base0, int n, double* scratch = nullptr,
                        long sScratch = 0) {
    const long ld = ws.Np, sM = ws.mat(), sW = scratch ? sScratch : ws.wstride();
    double* Wl = scratch ? scratch : ws.W;
    for (int s = 64; s < n; s *= 2) {
        const int nfull = n / (2 * s);                  // nodes with a full right child
        const int rem = n - nfull * 2 * s;              // tail: a partial node exists if rem > s
        for (int part = 0; part < 2; ++part) {
            int nodes, h2;
            long base;
            if (part == 0) { nodes = nfull; h2 = s; base = base0; }
            else { nodes = rem > s ? 1 : 0; h2 = rem - s; base = base0 + (long)nfull * 2 * s; }
            if (nodes == 0) continue;
            const long o11 = base * ld + base, o21 = (base + s) * ld + base, o22 = (base + s) * ld + base + s;
            const long snode = (long)2 * s * (ld + 1);
            GemmP t = gemm_base(cx);                    // W = L21 inv11
            t.A = ws.L + o21; t.lda = ld; t.a_mc = 0;
            t.B = ws.Inv + o11; t.ldb = ld; t.b_nc = 1; t.kflags = KB_GE_N;
            t.C = Wl; t.ldc = s;
            t.M = h2; t.N = s; t.K = s;
            t.zdiv = nodes; t.sA = snode; t.sB = snode; t.sC = (long)s * s; t.sA2 = sM; t.sB2 = sM; t.sC2 = sW;
            launch_gemm(t, nodes * ws.batch, stream);
            GemmP u = gemm_base(cx);                    // inv21 = -inv22 W
            u.A = ws.Inv + o22; u.lda = ld; u.a_mc = 0; u.kflags = KA_LE_M;
            u.B = Wl; u.ldb = s; u.b_nc = 1;
            u.C = ws.Inv + o21; u.ldc = ld;
            u.M = h2; u.N = s; u.K = h2; u.alpha = -1.0;
            u.zdiv = nodes; u.sA = snode; u.sB = (long)s * s; u.sC = snode; u.sA2 = sM; u.sB2 = sW; u.sC2 = sM;
            launch_gemm(u, nodes * ws.batch, stream);
        }
    }
}

static void trtri_levels(const Ctx& cx, Workspace& ws) { trtri_range(cx, ws, cx.stream, 0, ws.Np); }

// One node [L11 0; L21 L22] of the inverse tree above the segment level, split in its two products so
// that the first can run as soon as the left child is inverted: W = L21 inv11 (into the node's own
// slot `wo` of ws.W), later inv21 = -inv22 W.
static void trtri_node_w(const Ctx& cx, Workspace& ws, hipStream_t stream, long base, int s, int h2, long wo) {
    const long ld = ws.Np, sM = ws.mat();
    GemmP t = gemm_base(cx);
    t.A = ws.L + (base + s) * ld + base; t.lda = ld; t.sA = sM; t.a_mc = 0;
    t.B = ws.Inv + base * ld + base; t.ldb = ld; t.sB = sM; t.b_nc = 1; t.kflags = KB_GE_N;
    t.C = ws.W + wo; t.ldc = s; t.sC = ws.wstride();
    t.M = h2; t.N = s; t.K = s;
    launch_gemm(t, ws.batch, stream);
}
static void trtri_node_inv(const Ctx& cx, Workspace& ws, hipStream_t stream, long base, int s, int h2, long wo) {
    const long ld = ws.Np, sM = ws.mat();
    GemmP u = gemm_base(cx);
    u.A = ws.Inv + (base + s) * ld + base + s; u.lda = ld; u.sA = sM; u.a_mc = 0; u.kflags = KA_LE_M;
    u.B = ws.W + wo; u.ldb = s; u.sB = ws.wstride(); u.b_nc = 1;
    u.C = ws.Inv + (base + s) * ld + base; u.ldc = ld; u.sC = sM;
    u.M = h2; u.N = s; u.K = h2; u.alpha = -1.0;
    launch_gemm(u, ws.batch, stream);
}

// The part of the inverse that becomes computable when rows [seg0, seg1) are factored (seg0 a multiple of
// SEGR): the levels inside the segment, then, smallest first, the second product of every higher node
// whose right child ends at seg1 and the first product of every node whose left child ends there.
static void trtri_segment(const Ctx& cx, Workspace& ws, hipStream_t stream, int seg0, int seg1) {
    const int Np = ws.Np;
    trtri_range(cx, ws, stream, seg0, seg1 - seg0);
    long wo = ws.hw() * ws.hw();
    for (int s = SEGR; s < Np; s *= 2)
        for (long base = 0; base + s < Np; base += 2 * (long)s) {
            const int h2 = (int)std::min<long>(s, Np - base - s);
            if (base + s + h2 == seg1) trtri_node_inv(cx, ws, stream, base, s, h2, wo);
            wo += (long)h2 * s;
        }
    wo = ws.hw() * ws.hw();
    for (int s = SEGR; s < Np; s *= 2)
        for (long base = 0; base + s < Np; base += 2 * (long)s) {
            const int h2 = (int)std::min<long>(s, Np - base - s);
            if (base + s == seg1) trtri_node_w(cx, ws, stream, base, s, h2, wo);
            wo += (long)h2 * s;
        }
}

static void factor_blocked(const Ctx& cx, Workspace& ws, bool do_chol, int k0 = 0) {
    const int Np = ws.Np, nb = Np / 64;
    const long ld = Np, sM = ws.mat();
    if (!do_chol) {   // inverse only (gpmpc_set_factors): all diagonal blocks are independent
        hipLaunchKernelGGL(leaf64_kernel, dim3(nb, 1, ws.batch), dim3(256), 0, cx.stream, (const double*)ws.L, ws.L,
                           ws.Inv, ld, sM, 0, 0, ws.info, cx.crow_mode, 15);
        trtri_levels(cx, ws);
        return;
    }
    for (int k = k0; k < nb; ++k) {     // k0 > 0: block columns < k0 are already factored and applied (gpmpc_append)
        const int off = 64 * k, M = Np - off - 64;
        hipLaunchKernelGGL(leaf64_kernel, dim3(1, 1, ws.batch), dim3(256), 0, cx.stream, (const double*)ws.K, ws.L, ws.Inv,
                           ld, sM, off, 1, ws.info, cx.crow_mode, 15);
        if (M <= 0) break;
        const long o11 = (long)off * ld + off, o21 = (long)(off + 64) * ld + off, o22 = (long)(off + 64) * ld + off + 64;
        GemmP p = gemm_base(cx);                        // panel: L21 = A21 inv_kk^T
        p.A = ws.K + o21; p.lda = ld; p.sA = sM; p.a_mc = 0;
        p.B = ws.Inv + o11; p.ldb = ld; p.sB = sM; p.b_nc = 0; p.kflags = KB_LE_N;
        p.C = ws.L + o21; p.ldc = ld; p.sC = sM;
        p.M = M; p.N = 64; p.K = 64;
        launch_gemm(p, ws.batch, cx.stream);
        GemmP q = gemm_base(cx);                        // trailing update: A22 -= L21 L21^T (lower)
        q.A = ws.L + o21; q.lda = ld; q.sA = sM; q.a_mc = 0;
        q.B = ws.L + o21; q.ldb = ld; q.sB = sM; q.b_nc = 0;
        q.C = ws.K + o22; q.ldc = ld; q.sC = sM;
        q.M = M; q.N = M; q.K = 64; q.alpha = -1.0; q.beta = 1.0; q.lower = 1;
        launch_gemm(q, ws.batch, cx.stream);
    }
    if (k0 == 0) trtri_levels(cx, ws);
}

static long long* g_chain_trace = nullptr;   // developer aid: GPMPC_CHAIN_TRACE=<file> dumps the chain's time stamps

// Two-level execution of the chained factorisation (batches of matrices -- C3's six outputs -- and Np > 4096, where the
// trailing matrix does not fit the tile-owner workers' registers).  The plain flagged execution below updates the WHOLE
// trailing matrix after every 64-column panel: a K = 64 product reads and writes 16 bytes of C per 128 flops and is
// bound by that traffic (C3: 85 ms for 2.2e12 flop).  Here W block columns form a super-panel:
//     for each super-panel [k0, k1):   chain kernel for blocks k0 .. k1-1 (one launch), panel rows and the trailing
//                                      update INSIDE the super-panel's columns as flagged K = 64 launches (small);
//                                      then ONE product A22 -= L21 L21^T with K = 64 W on everything to the right.
// C traffic of the big updates falls by W and they run at the GEMM's MFMA rate; the inverse of a finished 512-row
// segment runs on the third queue while the big update occupies the second.
// Blocked super-panels (r04).  The first version carried EVERY row below a super-panel through its K = 64 in-panel
// steps: per step the panel product and a trailing update over (all rows below) x (the super-panel's remaining columns),
// flag-coupled to the chain -- at a batch of 16 matrices of 4096^2 each of those launches moves ~0.5 GB and takes 0.1-0.5 ms,
// eight of them pace every super-panel (2.4 of the 2.5 ms a super-panel took; profiles/r04_batchfit_trace_before.txt).  Now
// the in-panel steps stay INSIDE the super-panel's diagonal block (64 W rows: L2-resident, latency only), and the rows
// below follow in one product with the block's explicit inverse, which the panel-wise triangular inverse needs anyway:
//     chain + flagged K = 64 launches on rows / columns [r_i, r_i+1)          ->  L11 = chol(A11)
//     I_i = L11^-1 (level-batched, own scratch)                               ->  trtri_range
//     L21 = A21 I_i^T   (rows below, K = 64 W, triangular)                    ->  one MFMA-rate product
//     A22 -= L21 L21^T  (K = 64 W; look-ahead split as before)
// want_inverse = false (value-only NLL evaluations of the restart search): L and the diagonal blocks' inverses I_i only
// (ws.inv_panels = W); twolevel_inverse_all forms L^-1 from them later, for the matrices that turn out to need it.
// the chain kernel's publications as write-through stores + flag, no L2 write-back (wg_sync.hpp; GPMPC_CHAIN_WT=0: release fence)
static int g_handoff_wt = -1;   // gpmpc_set_tuning("handoff_write_through", 0 / 1): both kernels; -1: GPMPC_CHAIN_WT / GPMPC_WORKER_WT or default
static int chain_wt_publish() {
    static const int v = getenv("GPMPC_CHAIN_WT") ? atoi(getenv("GPMPC_CHAIN_WT")) : 1;
    return g_handoff_wt >= 0 ? g_handoff_wt : v;
}
static int worker_wt_publish() {   // the same for the tile-owner workers and the courier (GPMPC_WORKER_WT)
    static const int v = getenv("GPMPC_WORKER_WT") ? atoi(getenv("GPMPC_WORKER_WT")) : 1;
    return g_handoff_wt >= 0 ? g_handoff_wt : v;
}
// Block columns per super-panel (GPMPC_TWOLEVEL=<n> pins it; 0 / 1 = off).  A function of the matrix size ONLY (a matrix's bits
// must not depend on its batch).  r04-r05: 8 everywhere.  r06 sweep (profiles/r06_sweep_twolevel_width.txt): wider panels halve
// the share of the K = 64 W updates' C traffic and the number of chain launches, until the in-panel work inside the (64 W)^2
// diagonal block dominates; even widths only (128-row tiles): N = 8192, Ny = 6: 8 / 10 / 12 / 14 / 16 -> factor 44.2 / 42.7 / 42.7 /
// 42.8 / 45.2 ms; N = 4096, 64 restarts: 99.4 / 102.9 / 101.7 / 104.1 / 92.7 restarts/s.
static int twolevel_width(int Np) {
#ifdef GPMPC_EMULATED
    static const int w = getenv("GPMPC_TWOLEVEL") ? atoi(getenv("GPMPC_TWOLEVEL")) : 2;
    (void)Np;
    return w;
#else
    static const int w = getenv("GPMPC_TWOLEVEL") ? atoi(getenv("GPMPC_TWOLEVEL")) : -1;
    if (w >= 0) return w;
    const int nb = Np / 64;
    return nb > 64 ? 10 : nb >= 28 ? 14 : 8;
#endif
}

// One step of the right-looking blocked inversion by row panels (see factor_twolevel): panel P_i = block columns [k0, k1).
// `batch` matrices, optionally the subset zmap[0 .. batch) of the workspace.
static void twolevel_inverse_panel(const Ctx& cx, Workspace& ws, hipStream_t st, int k0, int k1, bool have_I, bool blocked, int batch,
                                   const int* zmap = nullptr) {
    const int Np = ws.Np;
    const long ld = Np, sM = ws.mat(), sW = ws.wstride();
    const int ri = 64 * k0, a = 64 * (k1 - k0), rn = 64 * k1, Mb = Np - rn;
    if (!have_I) trtri_range(cx, ws, st, ri, a, blocked ? ws.Wl : nullptr, blocked ? ws.wl_stride() : 0);   // I_i
    if (ri > 0) {
        GemmP u = gemm_base(cx);                                           // T = -I_i S_i, then back into Inv[P_i, < r_i]
        u.A = ws.Inv + (long)ri * ld + ri; u.lda = ld; u.sA = sM; u.a_mc = 0; u.kflags = KA_LE_M;
        u.B = ws.Inv + (long)ri * ld; u.ldb = ld; u.sB = sM; u.b_nc = 1;
        u.C = ws.W; u.ldc = ri; u.sC = sW;
        u.M = a; u.N = ri; u.K = a; u.alpha = -1.0; u.zmap = zmap;
        launch_gemm(u, batch, st);
        hipLaunchKernelGGL(copy_rect_kernel, dim3((ri / 2 + 255) / 256, a, batch), dim3(256), 0, st, (const double*)ws.W,
                           (long)ri, sW, ws.Inv + (long)ri * ld, ld, sM, ri, zmap);   // (ri is a multiple of 64: pairs)
    }
    if (Mb > 0) {
        GemmP t = gemm_base(cx);                                           // new columns of S: L[> P_i, P_i] I_i
        t.A = ws.L + (long)rn * ld + ri; t.lda = ld; t.sA = sM; t.a_mc = 0;
        t.B = ws.Inv + (long)ri * ld + ri; t.ldb = ld; t.sB = sM; t.b_nc = 1; t.kflags = KB_GE_N;
        t.C = ws.Inv + (long)rn * ld + ri; t.ldc = ld; t.sC = sM;
        t.M = Mb; t.N = a; t.K = a; t.zmap = zmap;
        launch_gemm(t, batch, st);
        if (ri > 0) {
            GemmP v = gemm_base(cx);                                       // S[> P_i, < r_i] += L[> P_i, P_i] X[P_i, < r_i]
            v.A = ws.L + (long)rn * ld + ri; v.lda = ld; v.sA = sM; v.a_mc = 0;
            v.B = ws.Inv + (long)ri * ld; v.ldb = ld; v.sB = sM; v.b_nc = 1;
            v.C = ws.Inv + (long)rn * ld; v.ldc = ld; v.sC = sM;
            v.M = Mb; v.N = ri; v.K = a; v.beta = 1.0; v.zmap = zmap;
            launch_gemm(v, batch, st);
        }
    }
}

// L^-1 of `batch` matrices (the subset zmap of the workspace, or its first `batch`) from L and the inverses of the diagonal
// blocks of W block columns each, which a value-only factorisation left in Inv: every panel step of the inversion above, one
// after the other on `st`.  Same launches on the same operands as the inversion that accompanies a full factorisation.
static void twolevel_inverse_all(const Ctx& cx, Workspace& ws, hipStream_t st, int W, int batch, const int* zmap) {
    const int nb = ws.Np / 64;
    const bool blocked = ws.Wl && (long)(32 * W) * (32 * W) <= ws.wl_stride();
    for (int k0 = 0; k0 < nb; k0 += W) twolevel_inverse_panel(cx, ws, st, k0, std::min(nb, k0 + W), true, blocked, batch, zmap);
}

static bool factor_twolevel(const Ctx& cx, Workspace& ws, int spin_limit, int W, bool want_inverse = true) {
    const int Np = ws.Np, nb = Np / 64, nf = chain_flag_count(nb);
    const long ld = Np, sM = ws.mat(), sW = ws.wstride();
    int* leafdone = ws.flags + 1;
    int* pan1 = ws.flags + 1 + nb;
    int* tdone = ws.flags + 1 + 2 * nb;
    static const bool blocked_env = !(getenv("GPMPC_TWOLEVEL_BLOCKED") && atoi(getenv("GPMPC_TWOLEVEL_BLOCKED")) == 0);
    const int nsp = (nb + W - 1) / W;
    const bool blocked = blocked_env && ws.Wl && (long)(32 * W) * (32 * W) <= ws.wl_stride() && cx.seg && cx.n_seg >= 5 * nsp + 4;
    // The inverse follows panel by panel on the third queue -- right-looking blocked inversion of the row panels
    // P_i = super-panel i: with S = sum over finished panels m of L[., P_m] X[P_m, .] accumulated IN the not yet final
    // rows of Inv,
    //     I_i = (L[P_i, P_i])^-1 (level-batched),    X[P_i, < r_i] = -I_i S[P_i, < r_i],
    //     S[> P_i, < r_{i+1}] += L[> P_i, P_i] X[P_i, < r_{i+1}]                  (K = 64 W products)
    // so every step only needs rows P_i of L -- final as soon as super-panel i is factored -- and after the last
    // super-panel just its own inverse and one 64 W-row product remain (the tree-shaped inverse left the two products of
    // its root, a third of the fit, for the end).  Needs a 64 W x Np scratch panel in ws.W and the event pool.
    const bool panel_inv = want_inverse && cx.aux && cx.seg && cx.n_seg >= 3 && (long)64 * W * Np <= sW;
    auto inverse_panel = [&](hipStream_t st, int k0, int k1, bool have_I) {
        twolevel_inverse_panel(cx, ws, st, k0, k1, have_I, blocked, ws.batch);
    };
    int ev = 0;                                                // event pool cursor
    int inv_done = 0;                                          // block columns whose inverse panel has been enqueued
    // Look-ahead: the K = 64 W update of super-panel s is split in A(s) = the NEXT super-panel's columns (second queue,
    // what the chain needs next) and B(s) = everything right of them (fourth queue, low priority), so that B(s) overlaps
    // the latency-bound factorisation of super-panel s+1.  Order on shared tiles: A(s) after B(s-1) (event), B(s) after
    // the panels of s (event) and after B(s-1) (queue order).
    static const bool lookahead_on = !(getenv("GPMPC_LOOKAHEAD") && atoi(getenv("GPMPC_LOOKAHEAD")) == 0);
    const bool lookahead = lookahead_on && cx.bulk && cx.seg && cx.n_seg >= 5 * nsp + 4;
    hipEvent_t evB_prev = nullptr;
    if (lookahead) {
        hipEventRecord(cx.join, cx.stream);                    // the fourth queue starts behind everything enqueued so far
        hipStreamWaitEvent(cx.bulk, cx.join, 0);
    }
    for (int k0 = 0; k0 < nb; k0 += W) {
        const int k1 = std::min(nb, k0 + W), k2 = std::min(nb, k1 + W);
        const int rows_end = blocked ? 64 * k1 : Np;           // in-panel steps: inside the diagonal block / every row below
        // the chain of this super-panel starts when the update of its columns (second queue) is complete
        hipEventRecord(cx.join, cx.side);
        hipStreamWaitEvent(cx.stream, cx.join, 0);
        hipLaunchKernelGGL(chol_chain_kernel, dim3(1, 1, ws.batch), dim3(256), CHAIN_LDS_BYTES, cx.stream, (const double*)ws.K,
                           ws.L, ws.Inv, ld, sM, nb, ws.flags, (long)nf, ws.info, cx.crow_mode, spin_limit, g_chain_trace, 0, k0,
                           k1, 3, chain_wt_publish());
        hipLaunchKernelGGL(flag_gate_kernel, dim3(ws.batch), dim3(64), 0, cx.side, ws.flags, (long)nf, 1 + k0, 1, -1, 0,
                           spin_limit);                       // bulk workgroups only once this chain launch is resident
        for (int k = k0; k < k1; ++k) {
            const int off = 64 * k;
            const long o11 = (long)off * ld + off;
            const bool last = k + 1 == k1;                     // the chain stops after this leaf: row k+1 is the panel product's
            const int r0 = off + (last ? 64 : 128), M2 = rows_end - r0;
            if (M2 > 0) {
                GemmP p = gemm_base(cx);                       // panel: L(i,k) = A(i,k) inv_kk^T
                p.A = ws.K + (long)r0 * ld + off; p.lda = ld; p.sA = sM; p.a_mc = 0;
                p.B = ws.Inv + o11; p.ldb = ld; p.sB = sM; p.b_nc = 0; p.kflags = KB_LE_N;
                p.C = ws.L + (long)r0 * ld + off; p.ldc = ld; p.sC = sM;
                p.M = M2; p.N = 64; p.K = 64;
                p.wait_flag = leafdone + k; p.err = ws.flags; p.spin_limit = spin_limit; p.sFlags = nf;
                launch_gemm(p, ws.batch, cx.side);
            }
            const int M1 = rows_end - off - 64, N1 = 64 * (k1 - k - 1);   // trailing update inside the super-panel's columns
            if (!last && M1 > 64) {
                const long o1 = (long)(off + 64) * ld;
                GemmP q = gemm_base(cx);
                q.A = ws.L + o1 + off; q.lda = ld; q.sA = sM; q.a_mc = 0;
                q.B = ws.L + o1 + off; q.ldb = ld; q.sB = sM; q.b_nc = 0;
                q.C = ws.K + o1 + off + 64; q.ldc = ld; q.sC = sM;
                q.M = M1; q.N = N1; q.K = 64; q.alpha = -1.0; q.beta = 1.0; q.lower = 1;
                q.wait_flag = pan1 + k; q.err = ws.flags; q.spin_limit = spin_limit; q.sFlags = nf;
                q.skip00 = 1; q.done_flags = tdone + 2 * k;
                launch_gemm(q, ws.batch, cx.side, 64);         // flags are defined on 64 x 64 tiles
            }
        }
        bool have_I = false;
        if (blocked && k1 < nb) {
            // the diagonal block is factored when the chain launch ends (its last leaf has no flagged consumer): its inverse,
            // then every row below in one product
            const int ri = 64 * k0, a = 64 * (k1 - k0), rn = 64 * k1;
            hipEventRecord(cx.seg[ev], cx.stream);
            hipStreamWaitEvent(cx.side, cx.seg[ev], 0);
            ++ev;
            trtri_range(cx, ws, cx.side, ri, a, ws.Wl, ws.wl_stride());                   // I_i
            GemmP p = gemm_base(cx);                                                       // L21 = A21 I_i^T
            p.A = ws.K + (long)rn * ld + ri; p.lda = ld; p.sA = sM; p.a_mc = 0;
            p.B = ws.Inv + (long)ri * ld + ri; p.ldb = ld; p.sB = sM; p.b_nc = 0; p.kflags = KB_LE_N;
            p.C = ws.L + (long)rn * ld + ri; p.ldc = ld; p.sC = sM;
            p.M = Np - rn; p.N = a; p.K = a;
            launch_gemm(p, ws.batch, cx.side);
            have_I = true;
        }
        // rows < 64 k1 of L are final: the inverse of this row panel goes to the third queue, next to the big update
        if (panel_inv && k1 < nb && ev + 2 < cx.n_seg) {
            hipEventRecord(cx.seg[ev], cx.side);
            hipStreamWaitEvent(cx.aux, cx.seg[ev], 0);
            ++ev;
            hipEventRecord(cx.seg[ev], cx.stream);            // (the leaf's own stores: the chain launch has to be complete)
            hipStreamWaitEvent(cx.aux, cx.seg[ev], 0);
            ++ev;
            if (inv_done < k0) inverse_panel(cx.aux, inv_done, k0, false);   // (panels skipped for want of events: as one)
            inverse_panel(cx.aux, k0, k1, have_I && inv_done <= k0);
            inv_done = k1;
        }
        if (k1 < nb) {                                         // A22 -= L21 L21^T, K = 64 (k1 - k0)
            const long r = 64L * k1, c0 = 64L * k0;
            GemmP g = gemm_base(cx);
            g.A = ws.L + r * ld + c0; g.lda = ld; g.sA = sM; g.a_mc = 0;
            g.B = ws.L + r * ld + c0; g.ldb = ld; g.sB = sM; g.b_nc = 0;
            g.C = ws.K + r * ld + r; g.ldc = ld; g.sC = sM;
            g.M = Np - (int)r; g.N = Np - (int)r; g.K = 64 * (k1 - k0); g.alpha = -1.0; g.beta = 1.0; g.lower = 1;
            if (!lookahead || k2 >= nb) {
                if (lookahead && evB_prev) hipStreamWaitEvent(cx.side, evB_prev, 0);
                launch_gemm(g, ws.batch, cx.side);
            } else {
                hipEvent_t evP = cx.seg[ev++], evB = cx.seg[ev++];
                hipEventRecord(evP, cx.side);                  // the panels of this super-panel are complete
                if (evB_prev) hipStreamWaitEvent(cx.side, evB_prev, 0);
                GemmP ga = g;                                  // A(s): columns of the next super-panel
                ga.N = 64 * (k2 - k1);
                launch_gemm(ga, ws.batch, cx.side);
                const long r2 = 64L * k2;                      // B(s): the rest, on the fourth queue
                GemmP gb = g;
                gb.A = ws.L + r2 * ld + c0;
                gb.B = ws.L + r2 * ld + c0;
                gb.C = ws.K + r2 * ld + r2;
                gb.M = Np - (int)r2; gb.N = Np - (int)r2;
                hipStreamWaitEvent(cx.bulk, evP, 0);
                // B(s) fills every CU for milliseconds, and the next super-panel's chain workgroups need a CU's whole LDS
                // each: launched into that, they waited until B(s) had nothing left to dispatch (r04 trace, 16 matrices of
                // 4096^2: 1.2 ms between the chain launch and its first leaf, per super-panel).  So B(s) starts only once that
                // chain is resident -- its first leaf is out -- i.e. behind A(s) instead of next to it; what it then overlaps
                // is the latency-bound part of super-panel s+1 (chain, diagonal-block inverse), which is the point.
                static const bool gate_bulk = !(getenv("GPMPC_GATE_BULK") && atoi(getenv("GPMPC_GATE_BULK")) == 0);
                if (gate_bulk)
                    hipLaunchKernelGGL(flag_gate_kernel, dim3(ws.batch), dim3(64), 0, cx.bulk, ws.flags, (long)nf, 1 + k1, 1, -1, 0,
                                       spin_limit);
                launch_gemm(gb, ws.batch, cx.bulk);
                hipEventRecord(evB, cx.bulk);
                evB_prev = evB;
            }
        }
    }
    hipEventRecord(cx.join, cx.side);
    hipStreamWaitEvent(cx.stream, cx.join, 0);
    if (lookahead) {
        hipEventRecord(cx.fork, cx.bulk);
        hipStreamWaitEvent(cx.stream, cx.fork, 0);
    }
    if (cx.aux && cx.seg) {
        hipEventRecord(cx.seg[cx.n_seg - 1], cx.aux);
        hipStreamWaitEvent(cx.stream, cx.seg[cx.n_seg - 1], 0);
    }
    if (!want_inverse) {
        // value only: the last diagonal block's inverse completes the set I_0 .. I_last (forward substitution by blocks)
        trtri_range(cx, ws, cx.stream, 64 * ((nb - 1) / W * W), Np - 64 * ((nb - 1) / W * W), ws.Wl, ws.wl_stride());
        ws.inv_panels = W;
        return true;
    }
    if (!panel_inv) { trtri_levels(cx, ws); return true; }
    inverse_panel(cx.stream, inv_done, nb, false);             // what is left: the last panel (or everything not handed over)
    return true;
}

// Chained factorisation: the sequential part of every panel step runs in ONE persistent workgroup
// (chol_chain_kernel, main queue) that keeps a CU to itself, the bulk -- panel rows >= k+2 and the
// trailing update -- in ordinary GEMM launches on the side queue; flags in ws.flags couple the two.
// Returns false if the path is unavailable (no side queue).  A time-out inside the kernels is reported
// through ws.flags[0] and handled by the caller (fallback to factor_blocked).

static bool factor_chain(const Ctx& cx, Workspace& ws, int spin_limit, bool flags_cleared = false) {
    if (!cx.side || ws.Np < 128) return false;
    const int Np = ws.Np, nb = Np / 64, nf = chain_flag_count(nb);
    const long ld = Np, sM = ws.mat();
    if (!flags_cleared) hipMemsetAsync(ws.flags, 0, (size_t)ws.batch * nf * sizeof(int), cx.stream);
    hipEventRecord(cx.fork, cx.stream);
    hipStreamWaitEvent(cx.side, cx.fork, 0);
    // bulk work as tile-owner workers: 7 of 8 CUs run one, the trailing matrix lives in their registers
    // A worker fills a CU (512 threads x ~250 VGPRs) and the chain needs an empty CU too.  Measured on MI355X
    // (start-time stamps of the workers): workgroups are dealt to the shader engines (8 CUs each) in a fixed
    // rotation and a workgroup that does not fit on "its" engine waits there even when CUs are free elsewhere
    // -- with 8 workers on the engine that also got the chain, the 8th started 234 ms late, after the others'
    // polls had timed out.  So: 7 workers per engine, nothing else in flight but the chain (one matrix only).
    const int ntiles = (nb - 1) * nb / 2 - 1;            // tiles kept in registers (chol_worker.hpp)
    int NW = ws.batch == 1 && !cx.no_workers ? cx.workers - cx.workers / 8 : 0;
    // (tuning aid) GPMPC_NW1=<n>: workgroups of the first worker launch instead of 7 per shader engine -- beyond that a
    // workgroup may have to wait for the chain's engine (see above): the hand-off time-out and its fallback catch that
    static const int nw1_env = getenv("GPMPC_NW1") ? atoi(getenv("GPMPC_NW1")) : 0;
    if (NW > 0 && nw1_env > 0) NW = std::min(nw1_env, cx.workers - 1);
    // the last workgroup of every worker launch is the chain's courier (chol_worker.hpp), no tile owner; GPMPC_COURIER=0:
    // tile owners only (r03 A/B on one box: factor 1.675 -> 1.630 ms at C2 with the courier)
    static const bool worker_courier_env = !(getenv("GPMPC_COURIER") && atoi(getenv("GPMPC_COURIER")) == 0);
    const bool worker_courier = g_worker_courier >= 0 ? g_worker_courier != 0 : worker_courier_env;   // (gpmpc_set_tuning("worker_courier", ..))
    const int ncour = worker_courier ? 1 : 0;
    const int worker_maxt = worker_courier ? WORKER_MAXT_COURIER : WORKER_MAXT;
    if (NW > ntiles + ncour) NW = ntiles + ncour;
    const bool use_workers = NW >= 1 + ncour && nb >= 3 && (ntiles + (NW - ncour) - 1) / (NW - ncour) <= worker_maxt;
    // what the workers do not take: two-level panels (GPMPC_TWOLEVEL=<block columns per super-panel>, 0/1 = off)
    const int twolevel_W = twolevel_width(Np);
    if (!use_workers && twolevel_W > 1 && nb >= 2 * twolevel_W && cx.aux && cx.seg) {
        static const bool verbose2 = getenv("GPMPC_VERBOSE") != nullptr;
        if (verbose2)
            fprintf(stderr, "gpmpc: factor Np=%d batch=%d: two-level panels of %d block columns\n", Np, ws.batch, twolevel_W);
        // (value only: L and the diagonal blocks' inverses suffice -- but only the blocked super-panels leave those behind)
        const int nsp = (nb + twolevel_W - 1) / twolevel_W;
        const bool can_skip = cx.value_only && ws.Wl && (long)(32 * twolevel_W) * (32 * twolevel_W) <= ws.wl_stride() &&
                              cx.n_seg >= 5 * nsp + 4 && !(getenv("GPMPC_TWOLEVEL_BLOCKED") && atoi(getenv("GPMPC_TWOLEVEL_BLOCKED")) == 0);
        return factor_twolevel(cx, ws, spin_limit, twolevel_W, !can_skip);
    }
    // Worker launches and the row-panel schedule of the inverse.  The workers run as up to three launches
    // (GPMPC_MAX_LAUNCHES), cut where the tree of the triangular inverse has its nodes on the right spine (Np = 4096:
    // blocks 0-31, 32-47, 48-63 with 224 / 96 / 32 workers: after half of the steps three quarters of the tiles are
    // finished, and so on; a fourth launch for blocks 56-63 was measured slower, 2.33 against 2.11 ms).  A launch i that has finished leaves rows P_i = [r_i, r_i+1) of L final, and the CUs the NEXT launch does
    // not need run -- behind a gate that waits until that launch is resident, its workgroups need whole CUs -- the
    // part of L^-1 that is computable by then.  With S_j = (L[P_j, <r] L^-1[<r, <r]) for a later panel P_j, kept as a
    // matrix of its own and grown panel by panel,
    //     I_i = (L[P_i, P_i])^-1 (level-batched, trtri_range),    L^-1[P_i, <r_i] = -I_i S_i,
    //     W_j = L[P_j, P_i] I_i,   S_j <- [S_j - W_j S_i | W_j]                                    for every j > i,
    // so that after the chain only the LAST panel's own inverse and ONE product -I S remain.
    // (History, N = 4096, factor time: two launches 2.22-2.24 ms, three 2.11; pieces gated on the chain's progress by
    //  polling kernels instead of launch boundaries were slower)
    int s_top = 64;                                         // rows of the left child of the inverse tree's root
    while (2 * s_top < Np) s_top *= 2;
    if (s_top >= 256 && Np - s_top < s_top / 4) s_top /= 2;   // (Np a little above a power of two: not "everything, then a sliver")
    static const bool split_ok = !(getenv("GPMPC_WORKER_SPLIT") && atoi(getenv("GPMPC_WORKER_SPLIT")) == 0);
    static const int max_launches = getenv("GPMPC_MAX_LAUNCHES") ? atoi(getenv("GPMPC_MAX_LAUNCHES")) : 3;
    static const int nw2_env = getenv("GPMPC_NW2") ? atoi(getenv("GPMPC_NW2")) : 0;   // (tuning aids)
    static const int nw3_env = getenv("GPMPC_NW3") ? atoi(getenv("GPMPC_NW3")) : 0;
    static const int nw4_env = getenv("GPMPC_NW4") ? atoi(getenv("GPMPC_NW4")) : 0;
    // second launch: 96 of 256 CUs, <= 6 tiles per worker at Np = 4096 (measured: 64 / 96 / 128 / 160 / 192 workers ->
    // 2.44 / 2.40 / 2.43 / 2.53 / 2.61 ms; with the DMA-staged workers 64 .. 160 are within 1 %)
    const int nw_rule[4] = {NW, nw2_env > 0 ? nw2_env : std::max(1, cx.workers * 3 / 8),
                            nw3_env > 0 ? nw3_env : std::max(1, cx.workers / 8), nw4_env > 0 ? nw4_env : std::max(1, cx.workers / 16)};
    int r[6] = {0, Np, Np, Np, Np, Np}, nws[5] = {NW, 0, 0, 0, 0};   // panel starts r[0..L], r[L] = Np; workers per launch
    int L = 1;
    long wofs[5] = {0, 0, 0, 0, 0};                         // S_j of panel j (1 <= j < L) inside ws.W, ld = r[j]
    if (use_workers && split_ok && cx.aux && cx.seg) {
        long wo = ws.hw() * ws.hw();
        static const int cut1 = getenv("GPMPC_CUT1") ? atoi(getenv("GPMPC_CUT1")) : 0;   // (tuning aid: block of the first cut)
        int start = (cut1 > 0 && 64 * cut1 < Np) ? 64 * cut1 : s_top;
        while (L < 4 && L < max_launches && L + 1 <= cx.n_seg - 1) {
            const int a = start - r[L - 1];                 // rows of the panel the new cut closes
            const int nbr = (Np - start) / 64, nt = (nbr - 1) * nbr / 2 - 1;
            int nw = nw_rule[L];
            if (nt > 0 && nw > nt + ncour) nw = nt + ncour;
            if (a < SEGR || nbr < 3 || nt < 1 || nw < 1 + ncour || (nt + (nw - ncour) - 1) / (nw - ncour) > worker_maxt) break;
            r[L] = start; nws[L] = nw; ++L;
            int nxt = 64;                                   // next cut: the left child of what remains
            while (2 * nxt < Np - start) nxt *= 2;
            if (nxt >= 256 && Np - start - nxt < nxt / 4) nxt /= 2;
            static const int cut2 = getenv("GPMPC_CUT2") ? atoi(getenv("GPMPC_CUT2")) : 0;   // (tuning aid: block of the second cut)
            if (L == 2 && cut2 > 0 && 64 * cut2 > start && 64 * cut2 < Np) nxt = 64 * cut2 - start;
            start += nxt;
            if (start >= Np) break;
        }
        r[L] = Np;
        // storage of S_j: panels 1 .. L-2 a_j x r_j, the last panel (Np - r[L-1]) x r[L-1]; drop cuts that do not fit
        for (;;) {
            long need = wo;
            for (int jj = 1; jj < L; ++jj) { wofs[jj] = need; need += (long)(r[jj + 1] - r[jj]) * r[jj]; }
            if (L == 1 || need <= ws.wstride()) break;
            --L; r[L] = Np;
        }
    }
    const bool split = L >= 2;
    auto product = [&](hipStream_t st, const double* A, long lda, int kfl, const double* B, long ldb, double* C, long ldc,
                       int M, int N, int K, double alpha, double beta) {   // C = alpha A B + beta C, A K-contiguous, B N-contiguous
        GemmP g = gemm_base(cx);
        g.A = A; g.lda = lda; g.sA = 0; g.a_mc = 0;
        g.B = B; g.ldb = ldb; g.sB = 0; g.b_nc = 1;
        g.C = C; g.ldc = ldc; g.sC = 0;
        g.kflags = kfl; g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
        launch_gemm(g, 1, st);
    };
    // inverse pipelined segment by segment behind the chain: next to GEMM launches only
    const bool pipelined = !use_workers && cx.aux && cx.seg && Np >= 4 * SEGR;
    int seg_done = 0;
    bool s_ready_recorded = false;
    // leafdone[k] together with pan1[k] (one release less on the chain's path; r01-r02 default next to workers) or on its own
    // right behind the leaf: since the courier and the workers' look-ahead start from inv_kk, the early publication wins
    // (r03 A/B: chain 1.310 -> 1.302 ms).  GPMPC_MERGE_PUBLISH=1: the old way.
    static const bool merge_publish = getenv("GPMPC_MERGE_PUBLISH") && atoi(getenv("GPMPC_MERGE_PUBLISH")) != 0;
    const int wt_publish = chain_wt_publish();
    static const int late_polls = getenv("GPMPC_LATE_POLLS") ? atoi(getenv("GPMPC_LATE_POLLS")) : 3;   // (tuning aid, chol_chain.hpp land())
    // leafdone[k] behind the panel row's products in prefetched steps (chol_chain.hpp); GPMPC_CHAIN_DEFER_PUBLISH=0: in front, as r04
    static const bool defer_publish = !(getenv("GPMPC_CHAIN_DEFER_PUBLISH") && atoi(getenv("GPMPC_CHAIN_DEFER_PUBLISH")) == 0);
    {   // the chain kernel ends with the last leaf, i.e. when L is complete: its duration is the Cholesky's
        ProfScope t(cx.prof, cx.stream, GPMPC_PH_CHAIN);
        hipLaunchKernelGGL(chol_chain_kernel, dim3(1, 1, ws.batch), dim3(256), CHAIN_LDS_BYTES, cx.stream, (const double*)ws.K,
                           ws.L, ws.Inv, ld, sM, nb, ws.flags, (long)nf, ws.info, cx.crow_mode, spin_limit, g_chain_trace,
                           use_workers && merge_publish ? 1 : 0, 0, -1, late_polls, wt_publish, defer_publish ? 1 : 0);
    }
    // Early status (below): ev_chain marks the END OF THE CHAIN KERNEL, so it is recorded here, while that kernel is the main
    // queue's last entry (GPMPC_EV_CHAIN_LATE=1, tuning aid: behind the join with the workers' queue as r04 had it).
    static const bool ev_chain_late = getenv("GPMPC_EV_CHAIN_LATE") && atoi(getenv("GPMPC_EV_CHAIN_LATE")) != 0;
    const bool early_status = cx.tail && cx.tail->want_early && use_workers && split;
    if (early_status && !ev_chain_late) hipEventRecord(TailState::get(cx.tail->ev_chain), cx.stream);
    static const bool verbose = getenv("GPMPC_VERBOSE") != nullptr;
    // (tuning aid) GPMPC_WORKER_LOOKAHEAD=0: the workers turn a panel tile into L(i,k) only at the top of step k
    static const bool worker_lookahead = !(getenv("GPMPC_WORKER_LOOKAHEAD") && atoi(getenv("GPMPC_WORKER_LOOKAHEAD")) == 0);
    if (verbose)
        fprintf(stderr, "gpmpc: factor Np=%d batch=%d: chain kernel + %s (%d launch%s), inverse %s\n", Np, ws.batch,
                use_workers ? "tile-owner workers" : "GEMM launches", use_workers ? L : 0, L == 1 ? "" : "es",
                split ? "by row panels behind the worker launches" : pipelined ? "pipelined" : "at the end");
    // Row panels of the inverse: the worker launches' panels; GPMPC_TAIL_CUTS="8,12" (tuning aid) cuts the LAST launch's panel
    // again at those blocks (counted from that launch's first block).  A panel that ends with a worker launch is handed
    // over by that launch's end (event), one inside the last launch by the chain's own flags: pan1[e] and colready[e] of
    // its last block column e say that rows <= e of L and the whole block column e (row e + 1: the chain's, below: the
    // workers') are final.  That leaves less to invert after the chain (two levels at 256 rows instead of four), but every
    // extra panel adds ~10 dependent launches across three queues to the side work, and that -- not the last panel's
    // inverse -- is what ends last: measured at C2 (r03) factor 1.65 ms without, 1.80 / 1.97 / 2.18 ms with 1 / 2 / 3 cuts.
    int pcut[12] = {0}, P = 0;                              // panel starts pcut[0..P], pcut[P] = Np
    for (int i = 0; i < L; ++i) pcut[P++] = r[i];
    if (split) {
        static const char* tail_env = getenv("GPMPC_TAIL_CUTS");
        const char* tc = tail_env ? tail_env : "";
        const int b0 = r[L - 1] / 64;
        while (*tc && P < 10) {
            char* e2 = nullptr;
            const long c = strtol(tc, &e2, 10);
            if (e2 == tc) break;
            tc = (*e2 == ',') ? e2 + 1 : e2;
            const int row = 64 * (b0 + (int)c);
            if (c > 0 && row > pcut[P - 1] && row + 128 <= Np) pcut[P++] = row;
        }
    }
    long sofs[12] = {0};                                    // S_j of panel j (1 <= j < P) inside ws.W, ld = pcut[j]
    for (;;) {
        pcut[P] = Np;
        long need = ws.hw() * ws.hw();
        for (int jj = 1; jj < P; ++jj) { sofs[jj] = need; need += (long)(pcut[jj + 1] - pcut[jj]) * pcut[jj]; }
        if (P == L || need <= ws.wstride()) break;
        P = L;                                              // (the refined S_j are a little larger than the coarse ones)
    }
    const int ev0 = L;                                      // events: 0 .. L-2 the launches, ev0 + i: I_i done
    if (use_workers) {
        for (int i = 0; i < L; ++i) {
            int* ready = i ? ws.flags + chain_ready_index(nb) + 2 * (i - 1) : nullptr;   // arrival counter + flag of launch i
            auto* worker = worker_courier ? chol_worker_kernel<WORKER_MAXT_COURIER, true> : chol_worker_kernel<WORKER_MAXT, false>;
            hipLaunchKernelGGL(worker, dim3(nws[i], 1, ws.batch), dim3(WORKER_THREADS), WORKER_LDS_BYTES, cx.side,
                               ws.K, ws.L, (const double*)ws.Inv, ld, sM, nb, ws.flags, (long)nf, cx.crow_mode, spin_limit,
                               r[i] / 64, i + 1 < L ? (r[i + 1] - r[i]) / 64 : nb, ready, g_chain_trace, worker_lookahead ? 1 : 0, worker_wt_publish());
            if (i + 1 < L) hipEventRecord(cx.seg[i], cx.side);        // launch i finished: rows P_i of L are final
        }
        const bool own_events = ev0 + P + 2 < cx.n_seg - 2;
        for (int i = 0; split && i + 1 < P; ++i) {
            const int ri = pcut[i], a = pcut[i + 1] - pcut[i];
            // I_i is eight latency-bound launches (~90 us) that need nothing of panel i-1's products, which still occupy
            // the inverse queue when launch i ends (r03 timeline: they ran until 0.19 ms after launch 2's end): they go to
            // a queue of their own from the second panel on (the low-priority one: a HIGH-priority queue for them made every
            // launch 5 x slower and the fit 2.67 ms), the products wait for them through an event.
            static const bool trtri_own_queue = !(getenv("GPMPC_TRTRI_QUEUE") && atoi(getenv("GPMPC_TRTRI_QUEUE")) == 0);
            hipStream_t tq = (i >= 1 && cx.bulk && trtri_own_queue && own_events) ? cx.bulk : cx.aux;
            // (all I_i share ONE level scratch: the previous one must be through with it -- an explicit event, not "it
            //  finished long ago": with several handles alive HIP multiplexes their streams onto a few hardware queues and
            //  the inverse queue of this handle can sit behind another handle's work for any length of time)
            if (tq != cx.aux && i >= 1) hipStreamWaitEvent(tq, cx.seg[ev0 + i - 1], 0);
            if (i + 1 < L) {
                // behind launch i + 1, once it is resident (its workgroups need whole CUs)
                hipStreamWaitEvent(tq, cx.seg[i], 0);
                hipLaunchKernelGGL(flag_gate_kernel, dim3(ws.batch), dim3(64), 0, tq, ws.flags, (long)nf,
                                   chain_ready_index(nb) + 2 * i + 1, 1, -1, 0, spin_limit);
            } else {
                const int e = pcut[i + 1] / 64 - 1;         // (e + 2 < nb: the cuts leave >= 128 rows)
                hipLaunchKernelGGL(flag_gate_kernel, dim3(ws.batch), dim3(64), 0, tq, ws.flags, (long)nf,
                                   chain_pan1_index(nb, e), 1, chain_colready_index(nb, e), 1, spin_limit);
            }
            trtri_range(cx, ws, tq, ri, a);                                        // I_i
            if (i + 2 == P) hipEventRecord(cx.seg[cx.n_seg - 2], tq);             // the side queues' last use of the level scratch
            hipEventRecord(cx.seg[ev0 + i], tq);                                   // I_i done
            if (tq != cx.aux) hipStreamWaitEvent(cx.aux, cx.seg[ev0 + i], 0);
            const double* Ii = ws.Inv + (long)ri * ld + ri;
            const double* Si = i ? ws.W + sofs[i] : nullptr;                       // a x ri
            for (int jj = i + 1; jj < P; ++jj) {
                const int rj = pcut[jj], hj = pcut[jj + 1] - pcut[jj];
                double* Sj = ws.W + sofs[jj];
                product(cx.aux, ws.L + (long)rj * ld + ri, ld, KB_GE_N, Ii, ld, Sj + ri, rj, hj, a, a, 1.0, 0.0);        // W_j
                if (i) product(cx.aux, Sj + ri, rj, 0, Si, ri, Sj, rj, hj, ri, a, -1.0, 1.0);                            // S_j -= W_j S_i
            }
            // (the last panel's S is complete here: the final product after the chain waits for THIS point, not for the
            //  rows of L^-1 that follow -- nobody reads them before the factorisation is over)
            if (i + 2 == P && own_events) { hipEventRecord(cx.seg[cx.n_seg - 1], cx.aux); s_ready_recorded = true; }
            if (i) product(cx.aux, Ii, ld, KA_LE_M, Si, ri, ws.Inv + (long)ri * ld, ld, a, ri, a, -1.0, 0.0);             // L^-1[P_i, <r_i]
        }
    } else {
        hipLaunchKernelGGL(chain_gate_kernel, dim3(ws.batch), dim3(64), 0, cx.side, ws.flags, (long)nf, spin_limit);
        int* leafdone = ws.flags + 1;
        int* pan1 = ws.flags + 1 + nb;
        int* tdone = ws.flags + 1 + 2 * nb;
        for (int k = 0; k + 1 < nb; ++k) {
            const int off = 64 * k;
            const long o11 = (long)off * ld + off;
            const int M2 = Np - off - 128;                       // panel rows >= k+2 (row k+1 is the chain's)
            if (M2 > 0) {
                const long o2 = (long)(off + 128) * ld + off;
                GemmP p = gemm_base(cx);
                p.A = ws.K + o2; p.lda = ld; p.sA = sM; p.a_mc = 0;
                p.B = ws.Inv + o11; p.ldb = ld; p.sB = sM; p.b_nc = 0; p.kflags = KB_LE_N;
                p.C = ws.L + o2; p.ldc = ld; p.sC = sM;
                p.M = M2; p.N = 64; p.K = 64;
                p.wait_flag = leafdone + k; p.err = ws.flags; p.spin_limit = spin_limit; p.sFlags = nf;
                launch_gemm(p, ws.batch, cx.side);
            }
            const int M1 = Np - off - 64;                        // trailing update from block k+1 on, minus tile (k+1,k+1)
            if (M1 > 64) {
                const long o1 = (long)(off + 64) * ld;
                GemmP q = gemm_base(cx);
                q.A = ws.L + o1 + off; q.lda = ld; q.sA = sM; q.a_mc = 0;
                q.B = ws.L + o1 + off; q.ldb = ld; q.sB = sM; q.b_nc = 0;
                q.C = ws.K + o1 + off + 64; q.ldc = ld; q.sC = sM;
                q.M = M1; q.N = M1; q.K = 64; q.alpha = -1.0; q.beta = 1.0; q.lower = 1;
                q.wait_flag = pan1 + k; q.err = ws.flags; q.spin_limit = spin_limit; q.sFlags = nf;
                q.skip00 = 1; q.done_flags = tdone + 2 * k;
                launch_gemm(q, ws.batch, cx.side, 64);           // flags are defined on 64 x 64 tiles
                // rows [.., 64(k+1)) are final once this update has consumed panel k: a finished segment goes to aux
                if (pipelined && (off + 64) % SEGR == 0 && seg_done < cx.n_seg - 1) {
                    hipEventRecord(cx.seg[seg_done], cx.side);
                    hipStreamWaitEvent(cx.aux, cx.seg[seg_done], 0);
                    trtri_segment(cx, ws, cx.aux, seg_done * SEGR, (seg_done + 1) * SEGR);
                    ++seg_done;
                }
            }
        }
    }
    hipEventRecord(cx.join, cx.side);
    hipStreamWaitEvent(cx.stream, cx.join, 0);
    if (early_status) {
        // Early status: the pivot word and the hand-off words are final when the chain kernel ends (the chain cannot
        // finish unless every worker launch became resident and delivered; nothing behind it polls).  The workers'
        // queue has nothing left but its last launch's drain by then: it waits for the chain kernel's event and copies
        // them, the caller waits on ev_info.  (The main queue meanwhile waits for the join above, i.e. for the workers.)
        TailState& ts = *cx.tail;
        if (ev_chain_late) hipEventRecord(TailState::get(ts.ev_chain), cx.stream);
        hipStreamWaitEvent(cx.side, ts.ev_chain, 0);
        hipMemcpyAsync(ts.pin_info, ws.info, ts.nb * sizeof(int), hipMemcpyDeviceToHost, cx.side);
        hipMemcpyAsync(ts.cerr, ws.flags, ts.nflag * sizeof(int), hipMemcpyDeviceToHost, cx.side);
        hipEventRecord(ts.ev_info, cx.side);
        ts.early_done = true;
    }
    if (split) {                                            // the last panel: its own inverse, then -I S
        // The inverse of the last panel needs nothing from the side queue but the level scratch, which that queue left
        // long ago (event recorded behind its last trtri_range); only the product waits for its S.  (Waiting for the
        // whole side queue first put its last product, which ends ~50 us after the chain, in front of these eight
        // latency-bound launches.)
        hipStreamWaitEvent(cx.stream, cx.seg[cx.n_seg - 2], 0);
        const int rl = pcut[P - 1], h = Np - rl;
        trtri_range(cx, ws, cx.stream, rl, h);
        if (!s_ready_recorded) hipEventRecord(cx.seg[cx.n_seg - 1], cx.aux);
        hipStreamWaitEvent(cx.stream, cx.seg[cx.n_seg - 1], 0);
        product(cx.stream, ws.Inv + (long)rl * ld + rl, ld, KA_LE_M, ws.W + sofs[P - 1], rl, ws.Inv + (long)rl * ld, ld,
                h, rl, h, -1.0, 0.0);
        if (s_ready_recorded) {                             // ... and for whatever the inverse queue still had to do
            hipEventRecord(cx.seg[ev0 + P], cx.aux);
            hipStreamWaitEvent(cx.stream, cx.seg[ev0 + P], 0);
        }
        return true;
    }
    if (!pipelined) { trtri_levels(cx, ws); return true; }
    // segments the side queue could not hand over (the last ones) are inverted after the chain, on the main queue
    hipEventRecord(cx.seg[cx.n_seg - 1], cx.aux);
    hipStreamWaitEvent(cx.stream, cx.seg[cx.n_seg - 1], 0);
    for (int sg = seg_done; sg * SEGR < Np; ++sg) trtri_segment(cx, ws, cx.stream, sg * SEGR, std::min(Np, (sg + 1) * SEGR));
    return true;
}

// w = L^-1 y and alpha = L^-T w as two HBM-bound matrix-vector products with the explicit inverse.
// y: [batch] vectors with stride sy.
// (ev_w: recorded behind w -- what the variance product's fused mean waits for instead of alpha)
// (solve_w: the first half alone -- the fit's tail forms w on the main queue, right behind the last product of L^-1, and the
//  rest of alpha on the workers' queue: a variance product with the fused mean follows w without changing queues)
static void solve_w(const Ctx& cx, Workspace& ws, const double* y, long sy) {
    const int Np = ws.Np;
    hipLaunchKernelGGL(gemv_rows_kernel, dim3(Np / 4, ws.batch), dim3(256), 0, cx.stream, ws.Inv, y, ws.w, Np, ws.mat(), sy,
                       (long)Np, 1);
}
static void solve_alpha(const Ctx& cx, Workspace& ws, const double* y, long sy, hipEvent_t ev_w = nullptr) {
    const int Np = ws.Np;
    solve_w(cx, ws, y, sy);
    if (ev_w) hipEventRecord(ev_w, cx.stream);
    const int chunks = (Np + GEMVT_ROWS - 1) / GEMVT_ROWS;         // partial sums go through the (now idle) inverse scratch
    hipLaunchKernelGGL(gemv_lowerT_part_kernel, dim3((Np + 127) / 128, chunks, ws.batch), dim3(256), 0, cx.stream, ws.Inv, ws.w, ws.W,
                       Np, ws.mat(), (long)Np, ws.wstride());
    hipLaunchKernelGGL(gemv_lowerT_finish_kernel, dim3((Np + 255) / 256, ws.batch), dim3(256), 0, cx.stream, ws.W, ws.alpha, Np,
                       chunks, ws.wstride(), (long)Np);
}

// w = L^-1 y by blocked forward substitution from L and the inverses I_i of its diagonal blocks of W block columns (which are
// the diagonal blocks of L^-1, so this works on a complete inverse as well as on what a value-only factorisation leaves):
//     w[P_i] = I_i (y[P_i] - L[P_i, < r_i] w[< r_i]),      two row-dot launches per panel (rowdot_kernel), `tmp` [batch][Np].
// The lock-step restart search takes every NLL value from this w, whether or not the point's L^-1 is ever formed.
static void fwd_subst(const Ctx& cx, Workspace& ws, int W, const double* y, long sy, double* tmp) {
    const int Np = ws.Np, nb = Np / 64;
    const long ld = Np, sM = ws.mat();
    for (int k0 = 0; k0 < nb; k0 += W) {
        const int ri = 64 * k0, a = 64 * (std::min(nb, k0 + W) - k0);
        const double* t = y;
        long st = sy;
        if (ri > 0) {
            hipLaunchKernelGGL(rowdot_kernel, dim3(a / 4, ws.batch), dim3(256), 0, cx.stream, (const double*)ws.L, ld, sM, ri, a, 0, ri, 0,
                               (const double*)ws.w, (long)Np, y, sy, tmp, (long)Np, -1.0);
            t = tmp;
            st = Np;
        }
        hipLaunchKernelGGL(rowdot_kernel, dim3(a / 4, ws.batch), dim3(256), 0, cx.stream, (const double*)ws.Inv, ld, sM, ri, a, ri, a, 1, t, st,
                           (const double*)nullptr, 0L, ws.w, (long)Np, 1.0);
    }
}

// alpha = L^-T w from the explicit inverse (the second half of solve_alpha), optionally for the subset zmap of the workspace
static void solve_alpha_from_w(const Ctx& cx, Workspace& ws, int batch, const int* zmap) {
    const int Np = ws.Np;
    const int chunks = (Np + GEMVT_ROWS - 1) / GEMVT_ROWS;
    hipLaunchKernelGGL(gemv_lowerT_part_kernel, dim3((Np + 127) / 128, chunks, batch), dim3(256), 0, cx.stream, ws.Inv, ws.w, ws.W,
                       Np, ws.mat(), (long)Np, ws.wstride(), zmap);
    hipLaunchKernelGGL(gemv_lowerT_finish_kernel, dim3((Np + 255) / 256, batch), dim3(256), 0, cx.stream, ws.W, ws.alpha, Np,
                       chunks, ws.wstride(), (long)Np, zmap);
}

// The device copy of a persistent product's tile lists (vargemm_persist.hpp), made once per shape and device and kept for
// the life of the process (a few hundred KB each).
struct SchedKey {
    int device, mode, tilesM, tilesN, batch, K, slots;
    bool operator<(const SchedKey& o) const {
        return std::tie(device, mode, tilesM, tilesN, batch, K, slots) < std::tie(o.device, o.mode, o.tilesM, o.tilesN, o.batch, o.K, o.slots);
    }
};
struct SchedEntry { VarSchedDev v; long stamp; };
static std::map<SchedKey, SchedEntry> g_sched_cache;
static std::mutex g_sched_mutex;
static long g_sched_clock = 0;
constexpr size_t SCHED_CACHE_MAX = 96;           // distinct (shape, subset size) schedules kept per process; the least recently used goes

static int get_schedule(int device, int mode, int tilesM, int tilesN, int batch, int K, int slots, VarSchedDev* out) {
    std::lock_guard<std::mutex> lk(g_sched_mutex);
    const SchedKey key{device, mode, tilesM, tilesN, batch, K, slots};
    auto it = g_sched_cache.find(key);
    if (it != g_sched_cache.end()) { it->second.stamp = ++g_sched_clock; *out = it->second.v; return GPMPC_OK; }
    if (g_sched_cache.size() >= SCHED_CACHE_MAX) {       // (a lock-step search asks for one schedule per subset size m)
        auto old_it = g_sched_cache.begin();
        for (auto jt = g_sched_cache.begin(); jt != g_sched_cache.end(); ++jt)
            if (jt->second.stamp < old_it->second.stamp) old_it = jt;
        {   // a launch that still walks the evicted lists must be through -- on the device the lists live on, which need not be
            // the current one
            int cur = 0;
            (void)hipGetDevice(&cur);
            if (old_it->first.device != cur) (void)hipSetDevice(old_it->first.device);
            (void)hipDeviceSynchronize();
            hipFree(old_it->second.v.list);
            hipFree(old_it->second.v.off);
            if (old_it->first.device != cur) (void)hipSetDevice(cur);
        }
        g_sched_cache.erase(old_it);
    }
    const VarSchedule s = mode == PG_VAR ? var_schedule(tilesM, tilesN, batch, K, slots) : xtx_schedule(tilesM, batch, K, slots);
    VarSchedDev v;
    v.mode = mode; v.tilesM = tilesM; v.tilesN = tilesN; v.batch = batch; v.K = K; v.slots = slots;
    HIPCHK(hipMalloc(&v.list, std::max<size_t>(1, s.list.size()) * sizeof(int)));
    HIPCHK(hipMalloc(&v.off, s.off.size() * sizeof(int)));
    HIPCHK(hipMemcpy(v.list, s.list.data(), s.list.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(v.off, s.off.data(), s.off.size() * sizeof(int), hipMemcpyHostToDevice));
    if (getenv("GPMPC_VERBOSE"))
        std::fprintf(stderr, "gpmpc: %s schedule %d x %d x %d tiles on %d slots: heaviest slot %.0f half slabs, mean %.1f (+%.2f %%), %d of %d tiles at home\n",
                     mode == PG_VAR ? "variance" : "K^-1", tilesM, tilesN, batch, slots, s.max_load, s.mean_load,
                     100.0 * (s.max_load / s.mean_load - 1.0), s.home, (int)s.list.size());
    g_sched_cache[key] = SchedEntry{v, ++g_sched_clock};
    *out = v;
    return GPMPC_OK;
}

// Lower triangle of K^-1 = L^-T L^-1 for the first n matrices of `ws` (a6, optimize.py:489-490).  X = L^-1 is copied
// transposed into ws.K (the factorisation consumed it; only blocks on and above the diagonal are written, which no
// factorisation reads), so that both operands of X^T X are contiguous along the contraction index; the product is one
// persistent launch over a static schedule when the mean slot load reaches twice the longest tile, the one-tile-per-
// workgroup DMA kernel (64-row tiles) below that.  Both sum every element over the same slabs in the same order with the
// same LDS image, so an element's bits do not depend on n or on which of the two ran (the lock-step restart search
// relies on that: api_train.inl).
// zmap / zhost (device / host copy, optional): the n matrices are zhost[0 .. n) of the workspace instead of its first n.
static int invk_lower(const Ctx& cx, Workspace& ws, int n, const int* zmap = nullptr, const int* zhost = nullptr) {
    const int Np = ws.Np;
    const long sM = ws.mat();
    hipLaunchKernelGGL(transpose_lower_kernel, dim3(Np / 64, Np / 64, n), dim3(256), 0, cx.stream, (const double*)ws.Inv, ws.K, Np, zmap);
    GemmP p = gemm_base(cx);
    p.A = ws.K; p.lda = Np; p.sA = sM; p.a_mc = 0;
    p.B = ws.K; p.ldb = Np; p.sB = sM; p.b_nc = 0;
    p.kflags = KA_GE_M | KB_GE_N;
    p.C = ws.InvK; p.ldc = Np; p.sC = sM;
    p.M = Np; p.N = Np; p.K = Np; p.lower = 1;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    static const int persist_env = getenv("GPMPC_VARGEMM_PERSIST") ? atoi(getenv("GPMPC_VARGEMM_PERSIST")) : 1;
    const int persist = g_vargemm_persist >= 0 ? g_vargemm_persist : persist_env;
    const int T = (Np + VAR_TILE - 1) / VAR_TILE, slots = 2 * g_cu_count[dev];
    double total = 0.0;
    for (int tm = 0; tm < T; ++tm) total += (double)(tm + 1) * (2 * ((Np - tm * VAR_TILE) / 16) + 1);
    total *= n;
    if (persist && gemm_dma_supported(p) && n < 256 && T < 4096 && (persist > 1 || total / slots >= 2.0 * (2 * (Np / 16) + 1))) {
        VarSchedDev sd;
        CHK(get_schedule(dev, PG_XTX, T, T, n, Np, slots, &sd));
        launch_persist_gemm<PG_XTX>(p, sd, cx.stream, zmap);
    } else if (zmap) {                                            // a few scattered matrices: one launch each
        for (int i = 0; i < n; ++i) {
            GemmP q = p;
            q.A = q.B = ws.K + zhost[i] * sM;
            q.C = ws.InvK + zhost[i] * sM;
            launch_gemm(q, 1, cx.stream, 64);
        }
    } else {
        launch_gemm(p, n, cx.stream, 64);
    }
    return GPMPC_OK;
}

// K^-1 = L^-T L^-1 (lower triangle by MFMA, then mirrored)
static int compute_invK(const Ctx& cx, Workspace& ws) {
    CHK(ws_need_invK(ws));
    CHK(invk_lower(cx, ws, ws.batch));
    hipLaunchKernelGGL(symmetrize_kernel, dim3(ws.Np / 64, ws.Np / 64, ws.batch), dim3(256), 0, cx.stream, ws.InvK,
                       ws.Np);
    return GPMPC_OK;
}

