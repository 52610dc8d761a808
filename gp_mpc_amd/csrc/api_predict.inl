// api_predict.inl -- part of gpmpc_api.hip (one translation unit; included in order, not compiled alone).
// Concern: predict: I/O staging, mean / variance / Jacobian chunks, moment methods, derivative outputs, GP.covar.
// ------------------------------------------------------------------------------------------------
// predict
// ------------------------------------------------------------------------------------------------
static int chunk_size(const gpmpc_gp* h) {
    const double budget = 2.0e9;  // bytes of KsT scratch
    long c = (long)(budget / (8.0 * h->Np * h->Ny));
    c = c / 64 * 64;
    if (c < 64) c = 64;
    if (c > 32768) c = 32768;
    return (int)c;
}

// Host-pointer calls with little data (an MPC's shooting nodes at the reference's model sizes): the inputs are staged in
// a pinned buffer and go up in one copy, every output is a slice of one device block and comes down in one copy.  With a
// pageable hipMemcpyAsync per array a 'ME' prediction at N = 200 took 71 us of which the kernels are 30
// (tools/gpu_small_latency.sh); packed it takes one upload, the launches, one download and one synchronisation.
constexpr size_t IO_PACK_DOUBLES = 32768;      // 256 KB
struct IoPack {
    gpmpc_gp* h;
    bool on = false;
    size_t nin = 0, n = 0;
    struct Out { double* host; size_t off, cnt; };
    std::vector<Out> outs;
    static size_t pad(size_t c) { return (c + 1) & ~(size_t)1; }     // slices stay 16-byte aligned
    // total: doubles of all inputs and outputs (each padded); false -> the caller copies array by array as before
    int begin(gpmpc_gp* hh, bool host, size_t total) {
        h = hh;
        on = host && total <= IO_PACK_DOUBLES;
        if (!on) return GPMPC_OK;
        if (!h->io_dev) HIPCHK(hipMalloc(&h->io_dev, IO_PACK_DOUBLES * sizeof(double)));
        if (!h->io_pin) HIPCHK(hipHostMalloc((void**)&h->io_pin, IO_PACK_DOUBLES * sizeof(double), hipHostMallocDefault));
        return GPMPC_OK;
    }
    const double* in(const double* src, size_t cnt) {                  // call for all inputs first, then upload()
        std::memcpy(h->io_pin + n, src, cnt * sizeof(double));
        const double* dptr = h->io_dev + n;
        n += pad(cnt);
        nin = n;
        return dptr;
    }
    int upload() {
        if (nin) HIPCHK(hipMemcpyAsync(h->io_dev, h->io_pin, nin * sizeof(double), hipMemcpyHostToDevice, h->stream));
        return GPMPC_OK;
    }
    double* out(double* host_dst, size_t cnt) {                        // device slice for an output (nullptr for a NULL output)
        if (!host_dst) return nullptr;
        outs.push_back({host_dst, n, cnt});
        double* dptr = h->io_dev + n;
        n += pad(cnt);
        return dptr;
    }
    int download() {                                                   // one copy, one synchronisation, scatter on the host
        if (n > nin) HIPCHK(hipMemcpyAsync(h->io_pin + nin, h->io_dev + nin, (n - nin) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (const Out& o : outs) std::memcpy(o.host, h->io_pin + o.off, o.cnt * sizeof(double));
        return GPMPC_OK;
    }
};

// keep_tail: the caller is the first prediction behind a fit and orders itself against the fit's tail and alpha (predict_chunk)
static int ensure_scratch(gpmpc_gp* h, int B, bool keep_tail = false) {
    h->tail.armed = false;      // whoever comes here is about to use the predict scratch on the main queue (TailState)
    if (!keep_tail) alpha_ready(h);
    const int need = round_up(B < chunk_size(h) ? B : chunk_size(h), 64);
    if (need <= h->Bcap) return GPMPC_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    hipFree(h->Z); hipFree(h->Sigma); hipFree(h->KsT); hipFree(h->part); hipFree(h->partm); hipFree(h->meanT);
    hipFree(h->mean); hipFree(h->var); hipFree(h->J); hipFree(h->cov); hipFree(h->UT); hipFree(h->VT);
    hipFree(h->sensH); hipFree(h->sensV); hipFree(h->ccpart);
    h->UT = h->VT = h->sensH = h->sensV = h->ccpart = h->partm = nullptr;
    h->Z = h->Sigma = h->KsT = h->part = h->meanT = h->mean = h->var = h->J = h->cov = nullptr;
    h->Bcap = 0;
    const size_t d = h->d, Ny = h->Ny, Np = h->Np, Bc = need;
    HIPCHK(hipMalloc(&h->Z, Bc * d * sizeof(double)));
    HIPCHK(hipMalloc(&h->Sigma, Bc * d * d * sizeof(double)));
    HIPCHK(hipMalloc(&h->KsT, Ny * Bc * Np * sizeof(double)));
    // (per-row-tile partial sums: 16-row tiles at most for a batch, one row per wave -- Np / 4 tiles -- for <= 8 points, padded to 32)
    HIPCHK(hipMalloc(&h->part, Ny * std::max((Np / 16) * Bc, (Np / 4) * (size_t)32) * sizeof(double)));
    HIPCHK(hipMalloc(&h->partm, Ny * (Np / VAR_TILE + 1) * Bc * sizeof(double)));
    HIPCHK(hipMalloc(&h->meanT, Ny * Bc * sizeof(double)));
    HIPCHK(hipMalloc(&h->mean, Bc * Ny * sizeof(double)));
    HIPCHK(hipMalloc(&h->var, Bc * Ny * sizeof(double)));
    HIPCHK(hipMalloc(&h->J, Bc * Ny * d * sizeof(double)));
    HIPCHK(hipMalloc(&h->cov, Bc * Ny * Ny * sizeof(double)));
    HIPCHK(hipMalloc(&h->ccpart, (size_t)CROSSCOV_CHUNKS * Ny * CROSSCOV_SMALL_B * (d + 1) * sizeof(double)));
    h->Bcap = need;
    return GPMPC_OK;
}

// One chunk (B <= Bcap) with device pointers: mean/var (either may be NULL), optional J.
// VT (optional, with dVar): also keep V^T = (L^-1 Ks)^T, [Ny][Bp][Np], for the sensitivities
// behind_tail: this is the first use of the predict scratch behind a gpmpc_fit that returned at the end of its chain
// kernel (TailState, api_core.inl): the last row panel of L^-1 (eight latency-bound level launches
// and one product, ~0.3 ms in which the chip is nearly idle) is still in flight on the main queue and alpha follows it
// on the workers' queue.  The cross-covariances need neither: they are formed on the low-priority queue NOW, next to the
// tail, by a grid of one workgroup per CU (the tail's small launches keep finding room), the variance product follows
// the tail directly on the main queue, and the mean Ks^T alpha is formed on the workers' queue behind alpha, next to
// the variance product.  Only with device pointers on the handle's own queue (the inputs are ready when the call is
// made) and without the Jacobian (its sums are fused with alpha into the cross-covariance kernel).
static int predict_chunk(gpmpc_gp* h, int B, const double* dZ, double* dMean, double* dVar, double* dJ, double* VT = nullptr,
                         bool behind_tail = false) {
    const Ctx cx = h->cx();
    const int Bp = round_up(B, 32), Np = h->Np, Ny = h->Ny;
    int tilesM = 0;
    static const int overlap_env = getenv("GPMPC_PREDICT_OVERLAP") ? atoi(getenv("GPMPC_PREDICT_OVERLAP")) : 1;
    const bool overlapped = behind_tail && overlap_env && dVar && !dJ && !VT && B > 64 && cx.bulk && cx.side &&
                            h->stream == h->own_stream;
    TailState& ts = h->tail;
    // Large batches whose variance product runs as the persistent kernel take the mean from its fused reduction
    // (sum_i V_ij w_i, w = L^-1 y: vargemm_persist.hpp) instead of from the cross-covariance kernel or a kernel of its own:
    // decided here, before the cross-covariances are formed (GPMPC_FUSED_MEAN=0: as before).
    static const int persist_env0 = getenv("GPMPC_VARGEMM_PERSIST") ? atoi(getenv("GPMPC_VARGEMM_PERSIST")) : 1;
    static const bool fused_mean_env = !(getenv("GPMPC_FUSED_MEAN") && atoi(getenv("GPMPC_FUSED_MEAN")) == 0);
    bool fused_mean = false;
    if (dVar && dMean && !dJ && !VT && B > 64 && fused_mean_env) {
        const int persist0 = g_vargemm_persist >= 0 ? g_vargemm_persist : persist_env0;
        const int tM = (Np + VAR_TILE - 1) / VAR_TILE, tN = (Bp + VAR_TILE - 1) / VAR_TILE, slots0 = 2 * g_cu_count[h->device];
        GemmP q = gemm_base(cx);
        q.A = h->ws.Inv; q.lda = Np; q.sA = (long)Np * Np; q.kflags = KA_LE_M;
        q.B = h->KsT; q.ldb = Np; q.sB = (long)Bp * Np;
        q.M = Np; q.N = Bp; q.K = Np;
        const int tile0 = g_gemm_force_tile ? g_gemm_force_tile : gemm_pick_tile(q, Ny);
        fused_mean = persist0 && tile0 == VAR_TILE && Ny < 256 && tM < 4096 && tN < 4096 && gemm_dma_supported(q) &&
                     (long)tM * tN * Ny >= (persist0 > 1 ? 1 : 2L * slots0);
    }
    if (overlapped) {
        // (GPMPC_CROSSCOV_WGS: workgroups of the throttled launch; 0 = one per block of test points, i.e. not throttled)
        // (r05: half a workgroup per CU -- the launch is longer, 0.31 against 0.22 ms, still ends with the tail, and takes less
        //  from the tail's latency-bound launches: step -11 ... -25 us on two boxes, profiles/r05_sweep_cuts_throttle.txt)
        static const int cc_wgs = getenv("GPMPC_CROSSCOV_WGS") ? atoi(getenv("GPMPC_CROSSCOV_WGS")) : std::max(1, g_cu_count[h->device] / 2);
        {
            // (gpmpc_fit_predict_mean_var enqueues this before the chain has ended: the launch waits for the chain's end)
            if (ts.fused_early && ts.ev_chain) hipStreamWaitEvent(cx.bulk, ts.ev_chain, 0);
            ProfScope t(&h->prof, cx.bulk, GPMPC_PH_CROSSCOV);
            launch_crosscov(cx.bulk, h->d, h->XT, h->ws.hyper, nullptr, dZ, h->KsT, h->meanT, nullptr, h->N, Np, B, Bp, Ny, nullptr,
                            1, cc_wgs);
        }
        hipEventRecord(TailState::get(ts.ev_ks), cx.bulk);
        hipStreamWaitEvent(cx.stream, ts.ev_ks, 0);
        if (dMean && fused_mean) {
            // the variance product needs w = L^-1 y for its fused mean: the first kernel of the pending alpha
            if (ts.alpha_pending) hipStreamWaitEvent(cx.stream, ts.ev_w, 0);
        } else if (dMean) {                             // behind alpha (same queue), next to the variance product
            hipStreamWaitEvent(cx.side, ts.ev_ks, 0);
            if (!ts.alpha_pending) {                    // (alpha was formed on the main queue: behind the tail then)
                hipEventRecord(TailState::get(ts.ev_tail), cx.stream);
                hipStreamWaitEvent(cx.side, ts.ev_tail, 0);
            }
            hipLaunchKernelGGL((mean_dot_kernel<CROSSCOV_JT>), dim3(Bp / CROSSCOV_JT, Ny), dim3(256), 0, cx.side, h->KsT,
                               h->ws.alpha, h->meanT, Np, Bp);
            hipEventRecord(TailState::get(ts.ev_mean), cx.side);
        }
        ++h->n_behind_tail;
    } else {
        alpha_ready(h);                                 // (a caller that kept the tail state and did not qualify after all)
        PhaseTimer t(h, GPMPC_PH_CROSSCOV);
        // few test points (an MPC's shooting nodes): cut the training points in chunks so that the launch fills the chip
        const int nch = (Bp <= CROSSCOV_SMALL_B && Np >= CROSSCOV_CHUNK_MIN_NP) ? CROSSCOV_CHUNKS : 1;
        launch_crosscov(cx.stream, h->d, h->XT, h->ws.hyper, fused_mean ? nullptr : h->ws.alpha, dZ, h->KsT, h->meanT, dJ, h->N, Np, B, Bp, Ny,
                        h->ccpart, nch);
    }
    // One point: a dedicated kernel streams L^-1 once at 5.6 TB/s (C3 size).  Measured at N = 8192, Ny = 6
    // (tools/bench_smallb.py), its multi-column versions fall off quickly (B = 2 / 4 / 8: 0.41 / 0.52 / 0.82 ms)
    // while the DMA-staged GEMM below does any B <= 32 in 0.30-0.32 ms: GPMPC_VARSMALL_MAX (default 1) is the switch.
    static const int varsmall_max = getenv("GPMPC_VARSMALL_MAX") ? atoi(getenv("GPMPC_VARSMALL_MAX")) : 1;
    if (dVar && !VT && B <= varsmall_max && B <= 8) {
        PhaseTimer t(h, GPMPC_PH_VARGEMM);   // stream L^-1 once (HBM-bound), no MFMA padding waste
        static const int rpw_env = getenv("GPMPC_VARSMALL_RPW") ? atoi(getenv("GPMPC_VARSMALL_RPW")) : 0;   // (tuning aid: rows per wave, 1 .. 8)
        const int rpw = (rpw_env >= 1 && rpw_env <= 8 && Np % (4 * rpw_env) == 0) ? rpw_env : var_small_rows_per_wave(Np, Ny, g_cu_count[h->device]);
        tilesM = Np / (4 * rpw);
        const dim3 grid(tilesM, Ny);
        if (B == 1) hipLaunchKernelGGL((var_small_kernel<1>), grid, dim3(256), 0, cx.stream, h->ws.Inv, h->KsT, h->part, Np, Bp, rpw);
        else if (B == 2) hipLaunchKernelGGL((var_small_kernel<2>), grid, dim3(256), 0, cx.stream, h->ws.Inv, h->KsT, h->part, Np, Bp, rpw);
        else if (B <= 4) hipLaunchKernelGGL((var_small_kernel<4>), grid, dim3(256), 0, cx.stream, h->ws.Inv, h->KsT, h->part, Np, Bp, rpw);
        else hipLaunchKernelGGL((var_small_kernel<8>), grid, dim3(256), 0, cx.stream, h->ws.Inv, h->KsT, h->part, Np, Bp, rpw);
    } else if (dVar && B <= 64) {
        // small batch (an MPC's Nt shooting nodes): tall-skinny tiles, a row tile x all columns per workgroup,
        // so that L^-1 is streamed once and the small Ks panel is shared through LDS.  The stream is what matters:
        // the DMA-staged kernel with a THREE-image ring and 64-row tiles (several workgroups per CU, each with two
        // slabs in flight) reaches 5.0 TB/s of L^-1 at N = 8192, Ny = 6, B <= 32 (0.32 ms; four / five images 0.33 /
        // 0.34, 32-row tiles 0.39, 128-row tiles 0.40) where the register-staged 32-row kernel managed 3.5 TB/s
        // (0.46 ms; 128 rows 0.52, 64 rows 0.53, 16 rows 0.56).  33-64 columns: 64 x 64 tiles, 0.54 against 0.70 ms.
        // (A no-LDS direct-fragment streaming kernel, which re-reads the Ks panel from L2 once per row tile, was
        //  slower still: 0.66 ms at B = 30.)  GPMPC_SMALLB_DMA=0 selects the register-staged kernels.
        PhaseTimer t(h, GPMPC_PH_VARGEMM);
        GemmP p = gemm_base(cx);
        p.A = h->ws.Inv; p.lda = Np; p.sA = (long)Np * Np; p.a_mc = 0; p.kflags = KA_LE_M;
        p.B = h->KsT; p.ldb = Np; p.sB = (long)Bp * Np; p.b_nc = 0;
        p.M = Np; p.N = Bp; p.K = Np;
        p.epi = EPI_COLSUMSQ; p.part = h->part; p.ldpart = Bp;
        p.Ct = VT; p.ldct = Np; p.sCt = (long)Bp * Np;
        static const bool smallb_dma = !(getenv("GPMPC_SMALLB_DMA") && atoi(getenv("GPMPC_SMALLB_DMA")) == 0);
        const bool dma = smallb_dma && gemm_dma_supported(p);
        const int tm_rows = dma ? 64 : Bp <= 32 ? 32 : 128;
        tilesM = (Np + tm_rows - 1) / tm_rows;
        p.sPart = (long)tilesM * Bp;
        if (dma && Bp <= 32) launch_gemm_dma<64, 32, 4, 1, 3, 4>(p, Ny, cx.stream, 1 << 30, 2);
        else if (dma) launch_gemm_dma<64, 64, 2, 2, 3, 4>(p, Ny, cx.stream, 1 << 30, 2);
        else if (Bp <= 32) launch_gemm_cfg<32, 32, 32, 2, 1>(p, Ny, cx.stream, 1 << 30, 2);
        else launch_gemm_cfg<128, 64, 16, 4, 2>(p, Ny, cx.stream, 1 << 30, 2);
    } else if (dVar) {
        PhaseTimer t(h, GPMPC_PH_VARGEMM);
        GemmP p = gemm_base(cx);  // V = L^-1 Ks, reduced to column sums of squares in the epilogue
        p.A = h->ws.Inv; p.lda = Np; p.sA = (long)Np * Np; p.a_mc = 0; p.kflags = KA_LE_M;
        p.B = h->KsT; p.ldb = Np; p.sB = (long)Bp * Np; p.b_nc = 0;
        p.M = Np; p.N = Bp; p.K = Np;
        p.epi = EPI_COLSUMSQ; p.part = h->part; p.ldpart = Bp;
        p.Ct = VT; p.ldct = Np; p.sCt = (long)Bp * Np;
        const int tile = g_gemm_force_tile ? g_gemm_force_tile : gemm_pick_tile(p, Ny);
        tilesM = (Np + tile - 1) / tile;
        p.sPart = (long)tilesM * Bp;
        // 128-row tiles and at least two of them per workgroup slot: one persistent launch over a static schedule
        // (vargemm_persist.hpp; GPMPC_VARGEMM_PERSIST=0 / tuning knob 'vargemm_persist' 0: the dispatcher's order)
        static const int persist_env = getenv("GPMPC_VARGEMM_PERSIST") ? atoi(getenv("GPMPC_VARGEMM_PERSIST")) : 1;
        const int persist = g_vargemm_persist >= 0 ? g_vargemm_persist : persist_env;
        const int slots = 2 * g_cu_count[h->device], tilesN = (Bp + VAR_TILE - 1) / VAR_TILE;
        if (persist && tile == VAR_TILE && !VT && Ny < 256 && tilesM < 4096 && tilesN < 4096 && gemm_dma_supported(p) &&
            (long)tilesM * tilesN * Ny >= (persist > 1 ? 1 : 2L * slots)) {
            VarSchedDev sd;
            CHK(get_schedule(h->device, PG_VAR, tilesM, tilesN, Ny, Np, slots, &sd));
            if (fused_mean) { p.wvec = h->ws.w; p.sWv = Np; p.partm = h->partm; }
            launch_persist_gemm<PG_VAR>(p, sd, cx.stream);
            ++h->n_var_persist;
        } else {
            if (fused_mean) return fail(GPMPC_EINVAL, "internal: fused mean without the persistent variance product");
            launch_gemm(p, Ny, cx.stream, tile);
        }
    }
    if (overlapped && dMean && !fused_mean) {
        hipStreamWaitEvent(cx.stream, ts.ev_mean, 0);
        ts.alpha_pending = false;                       // (the main queue is now ordered behind alpha as well)
    }
    {
        PhaseTimer t(h, GPMPC_PH_FINISH);
        hipLaunchKernelGGL(var_finish_kernel, dim3(B), dim3(256), 0, cx.stream, h->part, h->meanT,
                           h->ws.hyper, dMean, dVar, B, Bp, Ny, h->d, tilesM, fused_mean ? (const double*)h->partm : nullptr);
        if (h->mean_kind && h->mean_add && (dMean || dJ))   // build_gp(meanFunc=...): mean += m(z), gp_functions.py:131,135
            hipLaunchKernelGGL(mean_add_kernel, dim3((unsigned)(((long)B * Ny + 255) / 256)), dim3(256), 0, cx.stream, dZ, h->mpar,
                               dMean, dJ, (double*)nullptr, h->mean_kind, B, Ny, h->d);
    }
    HIPCHK(hipGetLastError());
    return GPMPC_OK;
}

#include "predict_em.inl"

// Generic driver: handles host/device pointer modes and chunking.  Outputs any of mean[B][Ny],
// var[B][Ny], J[B][Ny][d], cov[B][Ny][Ny] (cov per `method`).
static int predict_driver(gpmpc_gp* h, int method, int B, const double* Z, const double* Sigma, double* mean,
                          double* var, double* J, double* cov) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    if (!h->fitted) return fail(GPMPC_ENOTFIT, "model has no factors (call gpmpc_fit or gpmpc_set_factors)");
    if (B <= 0 || !Z) return fail(GPMPC_EINVAL, "bad B or NULL Z");
    if (method == GPMPC_OLD_TA && h->mean_kind)   // gp_functions.py:311: m(inputmean) has Nx entries there, Y[:, a] - m(...) does not conform
        return fail(GPMPC_EINVAL, "'old_TA' with a non-zero mean function raises in the reference (gp_functions.py:309-311); not served");
    const bool need_sigma = (method == GPMPC_TA || method == GPMPC_EM || method == GPMPC_OLD_TA);
    if (cov && need_sigma && !Sigma) return fail(GPMPC_EINVAL, "this method needs the input covariance Sigma");
    HIPCHK(hipSetDevice(h->device));
    const int d = h->d, Ny = h->Ny;
    const bool host = h->ptr_mode == GPMPC_PTR_HOST;
    // (ensure_scratch drops `armed`: only the first scratch user behind a fit may take the route next to the fit's tail)
    const bool behind_tail = h->tail.armed && !host && !cov && !J && var && B <= chunk_size(h) && B > 64;
    CHK(ensure_scratch(h, B, behind_tail));
    const bool moments = cov && (method == GPMPC_EM || method == GPMPC_OLD_ME || method == GPMPC_OLD_TA);
    if (moments && !h->have_invK) {
        PhaseTimer t(h, GPMPC_PH_INVK);
        CHK(compute_invK(h->cx(), h->ws));
        h->have_invK = true;
    }
    for (int b0 = 0; b0 < B; b0 += h->Bcap) {
        const int nb = (B - b0 < h->Bcap) ? B - b0 : h->Bcap;
        const double* dZ = Z + (size_t)b0 * d;
        const double* dS = Sigma ? Sigma + (size_t)b0 * d * d : nullptr;
        const bool up_sigma = dS && cov && need_sigma;
        const size_t cZ = (size_t)nb * d, cS = (size_t)nb * d * d, cM = (size_t)nb * Ny, cJ = cM * d, cC = cM * Ny;
        IoPack io;
        CHK(io.begin(h, host, IoPack::pad(cZ) + (up_sigma ? IoPack::pad(cS) : 0) + (mean ? IoPack::pad(cM) : 0) +
                                  (var ? IoPack::pad(cM) : 0) + (J ? IoPack::pad(cJ) : 0) + (cov ? IoPack::pad(cC) : 0)));
        double *oMean, *oVar, *oJ, *oCov;
        if (io.on) {
            dZ = io.in(dZ, cZ);
            if (up_sigma) dS = io.in(dS, cS);
            CHK(io.upload());
            oMean = io.out(mean ? mean + (size_t)b0 * Ny : nullptr, cM);
            oVar = io.out(var ? var + (size_t)b0 * Ny : nullptr, cM);
            oJ = io.out(J ? J + (size_t)b0 * Ny * d : nullptr, cJ);
            oCov = io.out(cov ? cov + (size_t)b0 * Ny * Ny : nullptr, cC);
        } else {
            if (host) {
                HIPCHK(hipMemcpyAsync(h->Z, dZ, cZ * sizeof(double), hipMemcpyHostToDevice, h->stream));
                dZ = h->Z;
                if (up_sigma) {
                    HIPCHK(hipMemcpyAsync(h->Sigma, dS, cS * sizeof(double), hipMemcpyHostToDevice, h->stream));
                    dS = h->Sigma;
                }
            }
            oMean = mean ? (host ? h->mean : mean + (size_t)b0 * Ny) : nullptr;
            oVar = var ? (host ? h->var : var + (size_t)b0 * Ny) : nullptr;
            oJ = J ? (host ? h->J : J + (size_t)b0 * Ny * d) : nullptr;
            oCov = cov ? (host ? h->cov : cov + (size_t)b0 * Ny * Ny) : nullptr;
        }
        if (moments) {
            CHK(predict_moments_chunk(h, method, nb, dZ, dS, oMean ? oMean : h->mean, oCov));
        } else {
            const bool ta = cov && method == GPMPC_TA;
            double* jbuf = oJ ? oJ : (ta ? h->J : nullptr);
            double* vbuf = oVar ? oVar : (cov ? h->var : nullptr);
            CHK(predict_chunk(h, nb, dZ, oMean, vbuf, jbuf, nullptr, behind_tail));
            if (cov) {
                PhaseTimer t(h, GPMPC_PH_FINISH);
                const long ne = (long)nb * Ny * Ny;
                hipLaunchKernelGGL(cov_assemble_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, h->stream,
                                   vbuf, jbuf, ta ? dS : (const double*)nullptr, oCov, nb, Ny, d);
            }
        }
        if (io.on) {
            CHK(io.download());
        } else if (host) {
            if (mean) HIPCHK(hipMemcpyAsync(mean + (size_t)b0 * Ny, h->mean, (size_t)nb * Ny * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            if (var) HIPCHK(hipMemcpyAsync(var + (size_t)b0 * Ny, h->var, (size_t)nb * Ny * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            if (J) HIPCHK(hipMemcpyAsync(J + (size_t)b0 * Ny * d, h->J, (size_t)nb * Ny * d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            if (cov) HIPCHK(hipMemcpyAsync(cov + (size_t)b0 * Ny * Ny, h->cov, (size_t)nb * Ny * Ny * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
    }
    HIPCHK(hipGetLastError());
    return GPMPC_OK;
}

extern "C" int gpmpc_predict_mean_var(gpmpc_gp* h, int B, const double* Z, double* mean, double* var) {
    if (!mean && !var) return fail(GPMPC_EINVAL, "both outputs NULL");
    return predict_driver(h, GPMPC_ME, B, Z, nullptr, mean, var, nullptr, nullptr);
}

// Fused fit + predict: the same results as gpmpc_fit followed by gpmpc_predict_mean_var, bit for bit, with the prediction's
// launches enqueued BEFORE the host waits for the factorisation's status words: the host's round trip between the two calls
// (status words, return, next call: ~60-120 us at C2) is off the device's path; the cross-covariances start at the end of the
// chain kernel (event), next to the inverse's tail, throttled, exactly as the two calls form them.  Measured at C2: -10 ... -30 us
// per step (what bounds the step behind the chain is the inverse's tail, 0.33 ms, not the host).  Forming the
// cross-covariances -- which depend on the hyper-parameters and the points only -- INSIDE the chain's window instead was built and
// measured in r06 (commit 81ae31b; docs/history_r06.md): 0.5 ms slower (a sixth concurrently active stream slows every
// dispatch-bound kernel 3-4 x; on the existing queues the launch delays the second panel's inverse), removed again.
// Fast path: device pointers, the handle's own queue, 64 < B <= one scratch chunk, no K^-1; anything else is the two calls.
extern "C" int gpmpc_fit_predict_mean_var(gpmpc_gp* h, const double* hyper, int want_invK, int* info, int B, const double* Z,
                                          double* mean, double* var) {
    if (!h || !hyper) return fail(GPMPC_EINVAL, "NULL handle/hyper");
    if (B <= 0 || !Z) return fail(GPMPC_EINVAL, "bad B or NULL Z");
    if (!mean && !var) return fail(GPMPC_EINVAL, "both outputs NULL");
    static const bool fused_env = !(getenv("GPMPC_FUSED_FIT_PREDICT") && atoi(getenv("GPMPC_FUSED_FIT_PREDICT")) == 0);
    const bool fast = fused_env && h->ptr_mode == GPMPC_PTR_DEVICE && !want_invK && var && B > 64 && B <= chunk_size(h) &&
                      h->stream == h->own_stream && h->side_stream && h->bulk_stream && !h->mean_kind;
    if (!fast) {
        CHK(gpmpc_fit(h, hyper, want_invK, info));
        return gpmpc_predict_mean_var(h, B, Z, mean, var);
    }
    HIPCHK(hipSetDevice(h->device));
    alpha_ready(h);
    CHK(ensure_scratch(h, B, true));
    TailState& ts = h->tail;
    const std::function<int()> fused = [&]() {
        // (behind_tail only if this attempt returned its status early: the fallback executions run the plain route)
        return predict_chunk(h, B, Z, mean, var, nullptr, nullptr, ts.early_done);
    };
    ts.fused_early = true;
    const int rc = fit_impl(h, hyper, 0, info, &fused);
    ts.fused_early = false;
    ++h->n_fused;
    return rc;
}

extern "C" int gpmpc_mean_jac(gpmpc_gp* h, int B, const double* Z, double* mean, double* J) {
    if (!J) return fail(GPMPC_EINVAL, "J is NULL");
    return predict_driver(h, GPMPC_ME, B, Z, nullptr, mean, nullptr, J, nullptr);
}

extern "C" int gpmpc_predict_jac(gpmpc_gp* h, int method, int B, const double* Z, const double* Sigma, double* mean,
                                 double* cov, double* J) {
    if (method != GPMPC_ME && method != GPMPC_TA) return fail(GPMPC_EINVAL, "gpmpc_predict_jac serves the 'ME' and 'TA' methods");
    if (!mean || !cov || !J) return fail(GPMPC_EINVAL, "mean/cov/J NULL");
    return predict_driver(h, method, B, Z, Sigma, mean, nullptr, J, cov);
}

// T-step propagation; U given (open loop) or generated on the device from the state-feedback law (Kz, k0, Kc).
extern "C" int gpmpc_predict_sens(gpmpc_gp* h, int B, const double* Z, double* mean, double* var, double* J, double* Hm,
                                  double* dvar) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    if (!h->fitted) return fail(GPMPC_ENOTFIT, "model has no factors (call gpmpc_fit or gpmpc_set_factors)");
    if (B <= 0 || !Z) return fail(GPMPC_EINVAL, "bad B or NULL Z");
    HIPCHK(hipSetDevice(h->device));
    CHK(ensure_scratch(h, B));
    const int d = h->d, Ny = h->Ny, Np = h->Np;
    const bool host = h->ptr_mode == GPMPC_PTR_HOST;
    const bool second = Hm || dvar;
    // (no K^-1 here: K^-1 ks = L^-T (L^-1 ks), and L^-1 ks is what the variance product forms anyway)
    if (second && !h->UT) HIPCHK(hipMalloc(&h->UT, (size_t)Ny * h->Bcap * Np * sizeof(double)));
    if (second && !h->VT) HIPCHK(hipMalloc(&h->VT, (size_t)Ny * h->Bcap * Np * sizeof(double)));
    if (second && !h->sensH) {
        HIPCHK(hipMalloc(&h->sensH, (size_t)h->Bcap * Ny * d * d * sizeof(double)));
        HIPCHK(hipMalloc(&h->sensV, (size_t)h->Bcap * Ny * d * sizeof(double)));
    }
    const Ctx cx = h->cx();
    for (int b0 = 0; b0 < B; b0 += h->Bcap) {
        const int nb = (B - b0 < h->Bcap) ? B - b0 : h->Bcap;
        const double* dZ = Z + (size_t)b0 * d;
        const size_t cZ = (size_t)nb * d, cM = (size_t)nb * Ny, cJ = cM * d, cH = cJ * d;
        IoPack io;
        CHK(io.begin(h, host, IoPack::pad(cZ) + (mean ? IoPack::pad(cM) : 0) + (var ? IoPack::pad(cM) : 0) + (J ? IoPack::pad(cJ) : 0) +
                                  (Hm ? IoPack::pad(cH) : 0) + (dvar ? IoPack::pad(cJ) : 0)));
        double *oMean, *oVar, *oJ, *oH, *oV;
        if (io.on) {
            dZ = io.in(dZ, cZ);
            CHK(io.upload());
            oMean = io.out(mean ? mean + (size_t)b0 * Ny : nullptr, cM);
            oVar = io.out(var ? var + (size_t)b0 * Ny : nullptr, cM);
            oJ = io.out(J ? J + (size_t)b0 * Ny * d : nullptr, cJ);
            oH = io.out(Hm ? Hm + (size_t)b0 * Ny * d * d : nullptr, cH);
            oV = io.out(dvar ? dvar + (size_t)b0 * Ny * d : nullptr, cJ);
            if (!oH) oH = h->sensH;
            if (!oV) oV = h->sensV;
        } else {
            if (host) {
                HIPCHK(hipMemcpyAsync(h->Z, dZ, cZ * sizeof(double), hipMemcpyHostToDevice, h->stream));
                dZ = h->Z;
            }
            oMean = mean ? (host ? h->mean : mean + (size_t)b0 * Ny) : nullptr;
            oVar = var ? (host ? h->var : var + (size_t)b0 * Ny) : nullptr;
            oJ = J ? (host ? h->J : J + (size_t)b0 * Ny * d) : nullptr;
            oH = host ? h->sensH : (Hm ? Hm + (size_t)b0 * Ny * d * d : h->sensH);
            oV = host ? h->sensV : (dvar ? dvar + (size_t)b0 * Ny * d : h->sensV);
        }
        CHK(predict_chunk(h, nb, dZ, oMean, second ? (oVar ? oVar : h->var) : oVar, oJ, second ? h->VT : nullptr));
        if (second) {
            const int Bp = round_up(nb, 32);            // the layout predict_chunk left in KsT and VT
            PhaseTimer t(h, GPMPC_PH_FINISH);
            // UT[j][:] = (L^-T v_j)^T = (K^-1 ks_j)^T: one more pass over the lower triangle of L^-1, half the bytes of K^-1
            GemmP p = gemm_base(cx);
            if (Bp <= 64) {
                // the streaming orientation of the variance product (rows of L^-T per workgroup, all columns): U = L^-T V,
                // written transposed by the sum-of-squares epilogue (its sums land in `part`, free again, and are not used)
                p.A = h->ws.Inv; p.lda = Np; p.sA = h->ws.mat(); p.a_mc = 1; p.kflags = KA_GE_M;
                p.B = h->VT; p.ldb = Np; p.sB = (long)Bp * Np; p.b_nc = 0;
                p.M = Np; p.N = Bp; p.K = Np;
                p.epi = EPI_COLSUMSQ; p.part = h->part; p.ldpart = Bp; p.sPart = (long)(Np / 64) * Bp;
                p.Ct = h->UT; p.ldct = Np; p.sCt = (long)Bp * Np;
            }
            if (Bp <= 64 && gemm_dma_supported(p)) {
                if (Bp <= 32) launch_gemm_dma<64, 32, 4, 1, 3, 4>(p, Ny, cx.stream, 1 << 30, 2);
                else launch_gemm_dma<64, 64, 2, 2, 3, 4>(p, Ny, cx.stream, 1 << 30, 2);
            } else {
                p = gemm_base(cx);
                p.A = h->VT; p.lda = Np; p.sA = (long)Bp * Np; p.a_mc = 0;
                p.B = h->ws.Inv; p.ldb = Np; p.sB = h->ws.mat(); p.b_nc = 1; p.kflags = KB_GE_N;
                p.C = h->UT; p.ldc = Np; p.sC = (long)Bp * Np;
                p.M = Bp; p.N = Np; p.K = Np;
                launch_gemm(p, Ny, cx.stream);
            }
            launch_sens(cx.stream, d, h->XT, dZ, h->ws.hyper, h->ws.alpha, h->KsT, h->UT, oH, oV, h->N, Np, nb, Bp, Ny);
            if (h->mean_kind == GPMPC_MEAN_POLYNOMIAL && h->mean_add)
                hipLaunchKernelGGL(mean_add_kernel, dim3((unsigned)(((long)nb * Ny + 255) / 256)), dim3(256), 0, cx.stream, dZ,
                                   h->mpar, (double*)nullptr, (double*)nullptr, oH, h->mean_kind, nb, Ny, d);
        }
        if (io.on) {
            CHK(io.download());
        } else if (host) {
            if (mean) HIPCHK(hipMemcpyAsync(mean + (size_t)b0 * Ny, h->mean, (size_t)nb * Ny * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            if (var) HIPCHK(hipMemcpyAsync(var + (size_t)b0 * Ny, h->var, (size_t)nb * Ny * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            if (J) HIPCHK(hipMemcpyAsync(J + (size_t)b0 * Ny * d, h->J, (size_t)nb * Ny * d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            if (Hm) HIPCHK(hipMemcpyAsync(Hm + (size_t)b0 * Ny * d * d, h->sensH, (size_t)nb * Ny * d * d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            if (dvar) HIPCHK(hipMemcpyAsync(dvar + (size_t)b0 * Ny * d, h->sensV, (size_t)nb * Ny * d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
    }
    HIPCHK(hipGetLastError());
    return GPMPC_OK;
}

// ---- 'EM' with derivative outputs: what a casadi Callback for GP.__predict needs when the MPC
// propagates with exact moments (gp_class.py:220-224): value and Jacobians w.r.t. the input mean and covariance.
extern "C" int gpmpc_predict_em_sens(gpmpc_gp* h, int B, const double* Z, const double* Sigma, double* mean, double* cov,
                                     double* dmean_dz, double* dmean_dS, double* dcov_dz, double* dcov_dS) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    if (!h->fitted) return fail(GPMPC_ENOTFIT, "model has no factors (call gpmpc_fit or gpmpc_set_factors)");
    if (B <= 0 || !Z || !Sigma) return fail(GPMPC_EINVAL, "bad B or NULL Z / Sigma");
    const int d = h->d, Ny = h->Ny, Np = h->Np, N = h->N;
    const int KD = em_depth(d), EM_OPS_ORD = em_ops_ord(KD), EM_NSS = em_nss(KD);   // cross-term depth 8 (d <= 8) or 16
    HIPCHK(hipSetDevice(h->device));
    CHK(ensure_scratch(h, 1));
    if (!h->have_invK) {
        PhaseTimer t(h, GPMPC_PH_INVK);
        CHK(compute_invK(h->cx(), h->ws));
        h->have_invK = true;
    }
    CHK(ensure_beta(h));
    const Ctx cx = h->cx();
    const bool host = h->ptr_mode == GPMPC_PTR_HOST;
    const int P = Ny * (Ny + 1) / 2, PO = Ny * Ny, tiles = Np / 64;
    const size_t per_in = (size_t)PO * ((size_t)EM_OPS_ORD * Np + (size_t)tiles * EM_NSS + EM_NSS) * sizeof(double);
    int Bc = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, ((size_t)512 << 20) / per_in));
    // one device block: [Z | Sigma | mean | cov | dm_dz | dm_dS | dc_dz | dc_dS | prep | ops | part | sums]
    const size_t nZ = (size_t)B * d, nS = (size_t)B * d * d, nM = (size_t)B * Ny, nC = (size_t)B * Ny * Ny;
    const size_t n1 = nM * d, n2 = nM * d * d, n3 = nC * d, n4 = nC * d * d;
    const size_t nPrep = (size_t)B * (Ny + P) * (d * d + 1);
    const size_t nOps = (size_t)Bc * PO * EM_OPS_ORD * Np, nPart = (size_t)Bc * PO * tiles * EM_NSS, nSum = (size_t)Bc * PO * EM_NSS;
    CHK(ensure_em_scratch(h, (long)((nZ + nS + nM + nC + n1 + n2 + n3 + n4 + nPrep + nOps + nPart + nSum) * sizeof(double)), true));
    double* buf = h->ems;
    double *bZ = buf, *bS = bZ + nZ, *bM = bS + nS, *bC = bM + nM, *b1 = bC + nC, *b2 = b1 + n1, *b3 = b2 + n2, *b4 = b3 + n3,
           *prep = b4 + n4, *ops = prep + nPrep, *part = ops + nOps, *sums = part + nPart;
    int rc = GPMPC_OK;
    auto run = [&]() -> int {
        const double *dZ = Z, *dS = Sigma;
        // few inputs (an MPC's nodes): [Z | Sigma] goes up and [mean .. dcov_dS] comes down through the pinned mirror, one copy each
        const size_t nOut = nM + nC + n1 + n2 + n3 + n4;
        IoPack io;
        CHK(io.begin(h, host, std::max(nZ + nS, nOut)));
        if (io.on) {
            std::memcpy(h->io_pin, Z, nZ * sizeof(double));
            std::memcpy(h->io_pin + nZ, Sigma, nS * sizeof(double));
            HIPCHK(hipMemcpyAsync(bZ, h->io_pin, (nZ + nS) * sizeof(double), hipMemcpyHostToDevice, h->stream));
            dZ = bZ; dS = bS;
        } else if (host) {
            HIPCHK(hipMemcpyAsync(bZ, Z, nZ * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(bS, Sigma, nS * sizeof(double), hipMemcpyHostToDevice, h->stream));
            dZ = bZ; dS = bS;
        }
        // device-pointer mode writes straight into the caller's arrays; NULL outputs land in the scratch block
        double* oM = (!host && mean) ? mean : bM;
        double* oC = (!host && cov) ? cov : bC;
        double* o1 = (!host && dmean_dz) ? dmean_dz : b1;
        double* o2 = (!host && dmean_dS) ? dmean_dS : b2;
        double* o3 = (!host && dcov_dz) ? dcov_dz : b3;
        double* o4 = (!host && dcov_dS) ? dcov_dS : b4;
        for (int b0 = 0; b0 < B; b0 += Bc) {
            const int nb = std::min(Bc, B - b0);
            // (the mean is an operand of d cov; the covariance itself -- the value kernels' pair sums -- only on request)
            CHK(predict_moments_chunk(h, GPMPC_EM, nb, dZ + (size_t)b0 * d, dS + (size_t)b0 * d * d, oM + (size_t)b0 * Ny,
                                      cov ? oC + (size_t)b0 * Ny * Ny : nullptr));
            PhaseTimer t(h, GPMPC_PH_EM);
            if (b0 == 0) {
                hipLaunchKernelGGL(em_prep_kernel, dim3((unsigned)(B * (Ny + P))), dim3(DMAX * GJ_LD), 0, cx.stream, h->ws.hyper, dS,
                                   prep, B, Ny, d);
            }
#define GPMPC_EM_SENS(KDV)                                                                                                        \
            hipLaunchKernelGGL((em_mean_sens_kernel<KDV>), dim3(Ny, nb), dim3(256), 0, cx.stream, h->XT, dZ, h->beta, prep,        \
                               o1 + (size_t)b0 * Ny * d, o2 + (size_t)b0 * Ny * d * d, N, Np, d, Ny, b0);                            \
            hipLaunchKernelGGL((em_operands_ordered_kernel<KDV>), dim3((Np + 255) / 256, PO, nb), dim3(256), 0, cx.stream, h->XT, dZ, \
                               h->ws.hyper, prep, h->beta, ops, N, Np, d, Ny, b0);                                                   \
            hipLaunchKernelGGL((em_pair_sens_kernel<false, KDV>), dim3(tiles, PO, nb), dim3(256), 0, cx.stream, ops, h->ws.InvK,    \
                               h->XT, dZ, part, N, Np, Ny, d, b0, cx.crow_mode);                                                     \
            hipLaunchKernelGGL((em_pair_sens_kernel<true, KDV>), dim3(tiles, PO, nb), dim3(256), 0, cx.stream, ops, h->ws.InvK,     \
                               h->XT, dZ, part, N, Np, Ny, d, b0, cx.crow_mode);                                                     \
            hipLaunchKernelGGL((em_sens_reduce_kernel<KDV>), dim3(PO, nb), dim3(256), 0, cx.stream, part, sums, Ny, tiles);          \
            hipLaunchKernelGGL((em_sens_finish_kernel<KDV>), dim3((unsigned)(nb * P)), dim3(DMAX * GJ_LD), 0, cx.stream, sums, prep, \
                               h->ws.hyper, dS, oM, o1 + (size_t)b0 * Ny * d, o2 + (size_t)b0 * Ny * d * d,                          \
                               o3 + (size_t)b0 * Ny * Ny * d, o4 + (size_t)b0 * Ny * Ny * d * d, nb, Ny, d, b0);
            if (KD == 8) { GPMPC_EM_SENS(8) } else { GPMPC_EM_SENS(16) }     // d = 9 .. 16: the 16-deep instantiation
#undef GPMPC_EM_SENS
            HIPCHK(hipGetLastError());
        }
        if (io.on) {
            // (the upload has been consumed: every kernel above is ordered behind it on the stream, and this copy behind them)
            HIPCHK(hipMemcpyAsync(h->io_pin, bM, nOut * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            auto take = [&](double* dst, const double* dev_src, size_t n) {
                if (dst) std::memcpy(dst, h->io_pin + (dev_src - bM), n * sizeof(double));
            };
            take(mean, bM, nM); take(cov, bC, nC); take(dmean_dz, b1, n1); take(dmean_dS, b2, n2); take(dcov_dz, b3, n3); take(dcov_dS, b4, n4);
        } else if (host) {
            auto down = [&](double* dst, const double* src, size_t n) {
                return dst ? hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, h->stream) : hipSuccess;
            };
            HIPCHK(down(mean, bM, nM));
            HIPCHK(down(cov, bC, nC));
            HIPCHK(down(dmean_dz, b1, n1));
            HIPCHK(down(dmean_dS, b2, n2));
            HIPCHK(down(dcov_dz, b3, n3));
            HIPCHK(down(dcov_dS, b4, n4));
        }
        HIPCHK(hipStreamSynchronize(h->stream));
        return GPMPC_OK;
    };
    rc = run();
    if (rc != GPMPC_OK) hipStreamSynchronize(h->stream);
    return rc;
}

extern "C" int gpmpc_predict(gpmpc_gp* h, int method, int B, const double* Z, const double* Sigma, double* mean,
                             double* cov) {
    if (method < GPMPC_ME || method > GPMPC_OLD_TA) return fail(GPMPC_EINVAL, "No GP method with code %d", method);
    if (!mean || !cov) return fail(GPMPC_EINVAL, "mean/cov NULL");
    return predict_driver(h, method, B, Z, Sigma, mean, nullptr, nullptr, cov);
}

// ------------------------------------------------------------------------------------------------
// a14 GP.covar: covar[a] = sf^2 - V^T V, V = L^-1 ks(X, Xnew)   (gp_class.py:353-381)
// ------------------------------------------------------------------------------------------------
extern "C" int gpmpc_covar(gpmpc_gp* h, int n, const double* Xnew, double* covar) {
    if (!h || n <= 0 || !Xnew || !covar) return fail(GPMPC_EINVAL, "bad arguments");
    if (!h->fitted) return fail(GPMPC_ENOTFIT, "model has no factors");
    if (n > chunk_size(h)) return fail(GPMPC_EINVAL, "covar: n=%d exceeds the single-chunk limit %d", n, chunk_size(h));
    HIPCHK(hipSetDevice(h->device));
    CHK(ensure_scratch(h, n));
    const Ctx cx = h->cx();
    const int Np = h->Np, Ny = h->Ny, d = h->d, Bp = round_up(n, 64);
    const bool host = h->ptr_mode == GPMPC_PTR_HOST;
    const double* dZ = Xnew;
    if (host) {
        HIPCHK(hipMemcpyAsync(h->Z, Xnew, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
        dZ = h->Z;
    }
    launch_crosscov(cx.stream, d, h->XT, h->ws.hyper, h->ws.alpha, dZ, h->KsT, h->meanT, nullptr, h->N, Np, n, Bp, Ny);
    double *VT = nullptr, *C = nullptr;
    HIPCHK(hipMalloc(&VT, (size_t)Ny * Bp * Np * sizeof(double)));
    HIPCHK(hipMalloc(&C, (size_t)Ny * Bp * Bp * sizeof(double)));
    GemmP p = gemm_base(cx);  // VT[j][i] = sum_k KsT[j][k] invL[i][k]
    p.A = h->KsT; p.lda = Np; p.sA = (long)Bp * Np; p.a_mc = 0;
    p.B = h->ws.Inv; p.ldb = Np; p.sB = (long)Np * Np; p.b_nc = 0; p.kflags = KB_LE_N;
    p.C = VT; p.ldc = Np; p.sC = (long)Bp * Np;
    p.M = Bp; p.N = Np; p.K = Np;
    launch_gemm(p, Ny, cx.stream);
    GemmP q = gemm_base(cx);  // C = -VT VT^T
    q.A = VT; q.lda = Np; q.sA = (long)Bp * Np; q.a_mc = 0;
    q.B = VT; q.ldb = Np; q.sB = (long)Bp * Np; q.b_nc = 0;
    q.C = C; q.ldc = Bp; q.sC = (long)Bp * Bp;
    q.M = Bp; q.N = Bp; q.K = Np; q.alpha = -1.0;
    launch_gemm(q, Ny, cx.stream);
    std::vector<double> tmp((size_t)Ny * Bp * Bp);
    HIPCHK(hipMemcpyAsync(tmp.data(), C, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    hipFree(VT);
    hipFree(C);
    std::vector<double> out((size_t)Ny * n * n);
    for (int a = 0; a < Ny; ++a) {
        const double sf = h->hyper[(size_t)a * h->nh() + d];
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) out[((size_t)a * n + i) * n + j] = sf * sf + tmp[((size_t)a * Bp + i) * Bp + j];
    }
    if (host) std::memcpy(covar, out.data(), out.size() * sizeof(double));
    else HIPCHK(hipMemcpy(covar, out.data(), out.size() * sizeof(double), hipMemcpyHostToDevice));
    return GPMPC_OK;
}

