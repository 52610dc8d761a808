// api_fit.inl -- part of gpmpc_api.hip (one translation unit; included in order, not compiled alone).
// Concern: fit with the one-shot jitter rule, append (extension of the factors), factor import / export.
// ------------------------------------------------------------------------------------------------
// fit
// ------------------------------------------------------------------------------------------------
// gram + factor on a workspace whose hyper/jitter buffers are already on the device.
static void gram_and_factor(gpmpc_gp* h, Workspace& ws, bool no_workers = false, bool value_only = false) {
    Ctx cx = h->cx();
    cx.no_workers = no_workers;
    cx.value_only = value_only;
    ws.inv_panels = 0;                                   // (the execution that skips L^-1 says so)
    {
        PhaseTimer t(h, GPMPC_PH_GRAM);
        // (the K build also clears the status words and the chain's hand-off flags: no fill kernels in between)
        launch_gram(cx.stream, dim3(ws.Np / 64, ws.Np / 64, ws.batch), h->d, h->XT, ws.hyper, ws.jitter, ws.K, h->N, ws.Np, 0,
                    ws.info, ws.batch, ws.flags, ws.batch * chain_flag_count(ws.Np / 64));
    }
    {
        PhaseTimer t(h, GPMPC_PH_FACTOR);
        if (!(h->chain_mode && factor_chain(cx, ws, h->spin_limit, true))) factor_blocked(cx, ws, true);
    }
}

// Runs gram+Cholesky with the reference's one-shot jitter rule (optimize.py:345-350).
// info_out[b]: 0 ok, 1 jitter applied, <0: -(first bad pivot) after jitter.
// `post` enqueues the work that consumes the factors (alpha, K^-1, the NLL terms).  It goes into the stream right
// after the copies of the status words and BEFORE the host waits for them -- the host waits on an event recorded
// between the two -- so the host's round trip (wake up, inspect, return to the caller, next launches: ~50 us)
// overlaps with that work instead of leaving the device idle.  If the attempt turns out to have failed (jitter rule,
// hand-off time-out) the next attempt overwrites what `post` produced.
// max_attempts / jit_init (the lock-step restart search): ONE attempt with the given jitter on every matrix -- the caller
// repeats only the matrices that failed (info_out[b] < 0), as a batch of their own, instead of the whole batch.
static int factor_with_jitter(gpmpc_gp* h, Workspace& ws, const double* hyper_host, int* info_out,
                              const std::function<void()>& post = std::function<void()>(), int max_attempts = 2,
                              double jit_init = 0.0, bool no_workers = false, bool value_only = false) {
    const int nb = ws.batch;
    std::vector<double> jit(nb, jit_init);
    std::vector<int> info(nb, 0), res(nb, 0);
    const size_t nflag = (size_t)nb * chain_flag_count(ws.Np / 64);
    if (!h->ev_info) HIPCHK(hipEventCreateWithFlags(&h->ev_info, hipEventDisableTiming));
    if (h->pin_ints < nb + nflag) {
        if (h->pin) hipHostFree(h->pin);
        h->pin = nullptr;
        HIPCHK(hipHostMalloc((void**)&h->pin, (nb + nflag) * sizeof(int), hipHostMallocDefault));
        h->pin_ints = nb + nflag;
    }
    int* pin_info = h->pin;
    int* cerr = h->pin + nb;
    HIPCHK(hipMemcpyAsync(ws.hyper, hyper_host, (size_t)nb * (h->d + 2) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    const int mode_configured = h->chain_mode;
    if (h->chain_parked > 0 && --h->chain_parked == 0) h->chain_strikes = 0;      // re-arm the chained path
    if (h->chain_parked > 0) h->chain_mode = 0;
    struct Restore { gpmpc_gp* h; int m; ~Restore() { h->chain_mode = m; } } restore{h, mode_configured};
    std::unique_lock<std::mutex> turn(g_factor_mutex[h->device], std::defer_lock);
    if (h->chain_mode) turn.lock();               // held until the status words are back: the chain kernel and the worker launches have
                                                  // ended (what needs the whole chip); with early status the inverse's tail -- plain
                                                  // launches that share the device like any others -- may still be in flight
    for (int attempt = 0; attempt < max_attempts; ++attempt) {
        HIPCHK(hipMemcpyAsync(ws.jitter, jit.data(), nb * sizeof(double), hipMemcpyHostToDevice, h->stream));
        // (TailState) with the tile-owner workers the status words come back when the chain kernel ends, on the workers'
        // queue: this call then returns with the tail of the inverse (and `post`) still in flight on the main queue
        static const bool early_status = !(getenv("GPMPC_EARLY_STATUS") && atoi(getenv("GPMPC_EARLY_STATUS")) == 0);
        h->tail.pin_info = pin_info; h->tail.cerr = cerr; h->tail.nflag = nflag; h->tail.nb = nb;
        h->tail.ev_info = h->ev_info;
        h->tail.want_early = early_status && !g_chain_trace;
        h->tail.early_done = false;
        gram_and_factor(h, ws, no_workers, value_only);
        HIPCHK(hipGetLastError());
        const bool check_chain = h->chain_mode && h->side_stream && ws.Np >= 128;
        if (!h->tail.early_done) {
            HIPCHK(hipMemcpyAsync(pin_info, ws.info, nb * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            if (check_chain) HIPCHK(hipMemcpyAsync(cerr, ws.flags, nflag * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipEventRecord(h->ev_info, h->stream));
        }
        static const bool post_early = !(getenv("GPMPC_POST_EARLY") && atoi(getenv("GPMPC_POST_EARLY")) == 0);
        if (post && post_early) post();
        HIPCHK(hipEventSynchronize(h->ev_info));
        if (post && !post_early) post();
        for (int b = 0; b < nb; ++b) info[b] = pin_info[b];
        if (g_chain_trace && h->chain_mode) {
            HIPCHK(hipStreamSynchronize(h->stream));
            const size_t cnt = 1 << 20;    // chain stamps first, worker stamps from entry 4096 on (chol_worker.hpp)
            std::vector<long long> tr(cnt);
            HIPCHK(hipMemcpy(tr.data(), g_chain_trace, cnt * sizeof(long long), hipMemcpyDeviceToHost));
            if (FILE* f = fopen(getenv("GPMPC_CHAIN_TRACE"), "wb")) { fwrite(tr.data(), sizeof(long long), cnt, f); fclose(f); }
        }
        if (check_chain) {   // did a hand-off of the chained factorisation time out?
            int bad = 0;
            for (int b = 0; b < nb; ++b)
                if (cerr[(size_t)b * chain_flag_count(ws.Np / 64)] != 0) bad = cerr[(size_t)b * chain_flag_count(ws.Np / 64)];
            if (bad) {
                fprintf(stderr, "gpmpc: chained factorisation timed out on a hand-off (code %d); using the single-queue path\n", bad);
                if (getenv("GPMPC_VERBOSE")) {
                    const int nbk = ws.Np / 64;
                    fprintf(stderr, "  worker progress (1 + 4k + phase; 0 = never started):");
                    for (int wq = 0; wq < 256; ++wq) fprintf(stderr, "%s%d", wq % 32 ? " " : "\n    ", cerr[1 + 7 * nbk + wq]);
                    fprintf(stderr, "\n");
                    int tmin = 0x7fffffff;
                    for (int wq = 0; wq < 256; ++wq)
                        if (cerr[1 + 7 * nbk + wq]) tmin = std::min(tmin, cerr[1 + 7 * nbk + 256 + wq]);
                    fprintf(stderr, "  worker start times (us after the first):");
                    for (int wq = 0; wq < 256; ++wq)
                        fprintf(stderr, "%s%d", wq % 32 ? " " : "\n    ", cerr[1 + 7 * nbk + wq] ? cerr[1 + 7 * nbk + 256 + wq] - tmin : -1);
                    fprintf(stderr, "\n");
                    for (int q = 0; q < 7; ++q) {
                        fprintf(stderr, "  flags[%d]:", q);
                        for (int k = 0; k < std::min(nbk, 12); ++k) fprintf(stderr, " %d", cerr[1 + q * nbk + k]);
                        fprintf(stderr, "\n");
                    }
                }
                ++h->n_timeouts;
                if (++h->chain_strikes >= gpmpc_gp::CHAIN_STRIKES) h->chain_parked = gpmpc_gp::CHAIN_REARM;
                h->chain_mode = 0;                  // for the rest of THIS call (restored on return)
                h->tail.want_early = false;
                h->tail.early_done = false;
                HIPCHK(hipStreamSynchronize(h->stream));
                HIPCHK(hipStreamSynchronize(h->side_stream));
                if (h->aux_stream) HIPCHK(hipStreamSynchronize(h->aux_stream));
                if (h->bulk_stream) HIPCHK(hipStreamSynchronize(h->bulk_stream));
                gram_and_factor(h, ws, no_workers, value_only);
                HIPCHK(hipMemcpyAsync(pin_info, ws.info, nb * sizeof(int), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(hipEventRecord(h->ev_info, h->stream));
                if (post) post();
                HIPCHK(hipEventSynchronize(h->ev_info));
                for (int b = 0; b < nb; ++b) info[b] = pin_info[b];
            } else {
                h->chain_strikes = 0;
            }
        }
        if (h->chain_mode) ++h->n_chained; else ++h->n_single;
        bool any = false;
        for (int b = 0; b < nb; ++b)
            if (info[b] != 0) {
                any = true;
                if (attempt + 1 < max_attempts) { jit[b] = 1e-8; res[b] = 1; }
                else res[b] = -info[b];
            } else if (attempt == 0 && jit_init > 0.0) {
                res[b] = 1;                                 // (factored with the caller's jitter)
            }
        if (!any) break;
    }
    int rc = GPMPC_OK;
    for (int b = 0; b < nb; ++b) {
        if (info_out) info_out[b] = res[b];
        if (res[b] < 0) rc = GPMPC_ENOTPD;
    }
    if (rc != GPMPC_OK) return fail(rc, "K is not positive definite even after adding 1e-8*I");
    return GPMPC_OK;
}

// gpmpc_fit, and the fit half of gpmpc_fit_predict_mean_var (api_predict.inl): `fused` (optional) is called inside the post
// hook, i.e. with the factors' consumers being enqueued and BEFORE the host waits for the status words; it enqueues the
// prediction behind the tail.  (If the attempt fails -- jitter rule, hand-off time-out -- the next attempt calls it again.)
static int fit_impl(gpmpc_gp* h, const double* hyper, int want_invK, int* info, const std::function<int()>* fused) {
    if (!h || !hyper) return fail(GPMPC_EINVAL, "NULL handle/hyper");
    HIPCHK(hipSetDevice(h->device));
    const int nh = h->nh();
    for (int a = 0; a < h->Ny; ++a)
        for (int k = 0; k < h->d + 2; ++k) {
            const double v = hyper[(size_t)a * nh + k];
            if (!(v == v) || (k < h->d && v == 0.0) || (k == h->d && v == 0.0))
                return fail(GPMPC_EINVAL, "hyper[%d][%d] = %g is not a usable SE-ARD parameter", a, k, v);
        }
    h->fitted = false;
    h->have_invK = false;
    h->have_beta = false;
    h->tail.armed = false;
    int post_rc = GPMPC_OK;
    std::vector<double> kpart;           // [Ny][d+2]; y - m(X) goes to h->Yc (optimize.py:285,494)
    CHK(upload_mean_and_residual(h, hyper, h->Ny, kpart, &h->mpar, h->Y, &h->Yc));
    alpha_ready(h);                      // (a previous fit's alpha launches on the workers' queue: ordered before this fit's)
    bool alpha_on_side = false;
    CHK(factor_with_jitter(h, h->ws, kpart.data(), info, [&]() {
        static const bool alpha_side_env = !(getenv("GPMPC_ALPHA_SIDE") && atoi(getenv("GPMPC_ALPHA_SIDE")) == 0);
        alpha_on_side = h->tail.early_done && alpha_side_env && !want_invK && h->side_stream;
        if (alpha_on_side) {
            // the fit returns at the end of the chain kernel.  w = L^-1 y follows the inverse's tail on the main queue (r06: it
            // used to cross to the workers' queue first, two event hand-overs in front of a variance product that needs it for
            // its fused mean), the rest of alpha goes to the workers' queue, so that what the caller enqueues next on the main
            // queue -- a variance product -- follows w directly.  GPMPC_W_MAIN=0: all of alpha on the workers' queue, as r05.
            static const bool w_main = !(getenv("GPMPC_W_MAIN") && atoi(getenv("GPMPC_W_MAIN")) == 0);
            Ctx cs = h->cx();
            cs.stream = h->side_stream;
            if (w_main) {
                ProfScope t(&h->prof, h->stream, GPMPC_PH_SOLVE);
                solve_w(h->cx(), h->ws, h->y_model(), h->Np);
                hipEventRecord(TailState::get(h->tail.ev_w), h->stream);
                hipStreamWaitEvent(h->side_stream, h->tail.ev_w, 0);
                solve_alpha_from_w(cs, h->ws, h->ws.batch, nullptr);
                t.end_on(h->side_stream);
            } else {
                hipEventRecord(TailState::get(h->tail.ev_tail), h->stream);
                hipStreamWaitEvent(h->side_stream, h->tail.ev_tail, 0);
                ProfScope t(&h->prof, h->side_stream, GPMPC_PH_SOLVE);
                solve_alpha(cs, h->ws, h->y_model(), h->Np, TailState::get(h->tail.ev_w));
            }
            hipEventRecord(TailState::get(h->tail.ev_alpha), h->side_stream);
        } else {
            PhaseTimer t(h, GPMPC_PH_SOLVE);
            solve_alpha(h->cx(), h->ws, h->y_model(), h->Np);
        }
        if (want_invK) {
            PhaseTimer t(h, GPMPC_PH_INVK);
            post_rc = compute_invK(h->cx(), h->ws);
        }
        if (fused && post_rc == GPMPC_OK) {
            h->tail.alpha_pending = alpha_on_side;      // (what the prediction orders itself against)
            post_rc = (*fused)();
        }
    }));
    CHK(post_rc);
    if (want_invK) h->have_invK = true;
    HIPCHK(hipGetLastError());
    h->hyper.assign(hyper, hyper + (size_t)h->Ny * nh);
    h->fitted = true;
    // the first large prediction behind this fit may start next to the inverse's tail (predict_chunk)
    h->tail.alpha_pending = alpha_on_side;
    h->tail.armed = h->tail.early_done && !want_invK && !fused;
    return GPMPC_OK;
}

extern "C" int gpmpc_fit(gpmpc_gp* h, const double* hyper, int want_invK, int* info) {
    return fit_impl(h, hyper, want_invK, info, nullptr);
}

// ---- data update: a15 (GP.update_data_all gp_class.py:474-550 = append + full recomputation with the
// existing hyper-parameters) as a rank-n extension of the factors.
// With R0 = 64 floor(N/64) the rows < R0 of L and L^-1 do not change.  For the strip of m = Np' - R0 rows
// below (the last partial block of old points, the new points, padding):
//     K' rows >= R0 from the K build;   L21 = K21 inv11^T;   S = K22 - L21 L21^T;   L22 = chol(S) (blocked);
//     inv22 = L22^-1;   inv21 = -inv22 (L21 inv11)
// i.e. four GEMMs with K = R0 plus a factorisation of m rows -- O(N^2 m) instead of O(N^3).
static void free_predict_scratch(gpmpc_gp* h) {
    hipFree(h->Z); hipFree(h->Sigma); hipFree(h->KsT); hipFree(h->part); hipFree(h->partm); hipFree(h->meanT);
    hipFree(h->mean); hipFree(h->var); hipFree(h->J); hipFree(h->cov); hipFree(h->UT); hipFree(h->VT);
    hipFree(h->sensH); hipFree(h->sensV); hipFree(h->em); hipFree(h->ems); hipFree(h->beta); hipFree(h->gradPartial); hipFree(h->gradOut);
    hipFree(h->ccpart); hipFree(h->Yc); hipFree(h->tYc);
    h->Yc = h->tYc = nullptr;
    h->partm = nullptr;
    h->Z = h->Sigma = h->KsT = h->part = h->meanT = h->mean = h->var = h->J = h->cov = h->UT = h->VT = nullptr;
    h->sensH = h->sensV = h->em = h->ems = h->beta = h->gradPartial = h->gradOut = h->ccpart = nullptr;
    h->Bcap = 0;
    h->emBytes = h->emsBytes = 0;
    h->have_beta = false;
    ws_free(h->tws);
    ws_free(h->bws);
    hipFree(h->bYc); hipFree(h->bmpar); hipFree(h->bgradPartial); hipFree(h->bgradOut);
    h->bYc = h->bmpar = h->bgradPartial = h->bgradOut = nullptr;
}

// y - m(X) of the model's current data and stored mean parameters (after the data changed)
static int refresh_residual(gpmpc_gp* h) {
    if (!h->mean_kind) return GPMPC_OK;
    std::vector<double> unused;
    return upload_mean_and_residual(h, h->hyper.data(), h->Ny, unused, &h->mpar, h->Y, &h->Yc);
}

extern "C" int gpmpc_append(gpmpc_gp* h, int n, const double* Xnew, const double* Ynew, int* info) {
    if (!h || n <= 0 || !Xnew || !Ynew) return fail(GPMPC_EINVAL, "NULL handle/data or n <= 0");
    if (!h->fitted) return fail(GPMPC_ENOTFIT, "model has no factors (call gpmpc_fit or gpmpc_set_factors)");
    HIPCHK(hipSetDevice(h->device));
    alpha_ready(h);
    h->tail.armed = false;
    HIPCHK(hipStreamSynchronize(h->stream));
    const int N0 = h->N, N1 = N0 + n, d = h->d, Ny = h->Ny, Np0 = h->Np, Np1 = round_up(N1, 64);
    const int R0 = (N0 / 64) * 64, m = Np1 - R0;
    // new data buffers: old points back from the device, new ones appended
    std::vector<double> xt0((size_t)d * Np0), yt0((size_t)Ny * Np0), xt((size_t)d * Np1, 0.0), yt((size_t)Ny * Np1, 0.0);
    HIPCHK(hipMemcpy(xt0.data(), h->XT, xt0.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(yt0.data(), h->Y, yt0.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int k = 0; k < d; ++k) {
        std::memcpy(&xt[(size_t)k * Np1], &xt0[(size_t)k * Np0], N0 * sizeof(double));
        for (int i = 0; i < n; ++i) xt[(size_t)k * Np1 + N0 + i] = Xnew[(size_t)i * d + k];
    }
    for (int a = 0; a < Ny; ++a) {
        std::memcpy(&yt[(size_t)a * Np1], &yt0[(size_t)a * Np0], N0 * sizeof(double));
        for (int i = 0; i < n; ++i) yt[(size_t)a * Np1 + N0 + i] = Ynew[(size_t)i * Ny + a];
    }
    double *XT1 = nullptr, *Y1 = nullptr;
    HIPCHK(hipMalloc(&XT1, xt.size() * sizeof(double)));
    HIPCHK(hipMalloc(&Y1, yt.size() * sizeof(double)));
    HIPCHK(hipMemcpy(XT1, xt.data(), xt.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(Y1, yt.data(), yt.size() * sizeof(double), hipMemcpyHostToDevice));
    Workspace ws1;
    int rc = ws_alloc(ws1, Ny, Np1, d);
    if (rc != GPMPC_OK) { hipFree(XT1); hipFree(Y1); return rc; }
    HIPCHK(hipMemcpy(ws1.hyper, h->ws.hyper, (size_t)Ny * (d + 2) * sizeof(double), hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(ws1.jitter, h->ws.jitter, (size_t)Ny * sizeof(double), hipMemcpyDeviceToDevice));
    const long slot_cap = ws1.wstride() - ws1.hw() * ws1.hw();
    const bool strip = R0 >= 64 && m <= Np1 / 4 && (long)m * R0 <= slot_cap;
    h->nll_last_a = -1;                                     // (the training workspace's factors belong to the old data)
    auto install = [&]() {                                  // the handle takes the new data set
        hipFree(h->XT); hipFree(h->Y);
        ws_free(h->ws);
        free_predict_scratch(h);
        h->XT = XT1; h->Y = Y1; h->ws = ws1;
        h->N = N1; h->Np = Np1;
        h->have_invK = false;
        const size_t need = seg_event_count(Np1);
        while (h->seg_events.size() < need) {
            hipEvent_t e;
            hipEventCreateWithFlags(&e, hipEventDisableTiming);
            h->seg_events.push_back(e);
        }
    };
    if (!strip) {                                           // too many new rows for the update to pay: plain refit
        // the handle takes the new data set for the duration of the fit; if K turns out not to be positive definite
        // the old model (data, factors, K^-1 state) is put back, as the header promises
        double *XT0 = h->XT, *Y0 = h->Y;
        Workspace ws0 = h->ws;
        const bool invK0 = h->have_invK;
        const std::vector<double> hy = h->hyper;
        free_predict_scratch(h);
        h->XT = XT1; h->Y = Y1; h->ws = ws1;
        h->N = N1; h->Np = Np1;
        const size_t need = seg_event_count(Np1);
        while (h->seg_events.size() < need) {
            hipEvent_t e;
            hipEventCreateWithFlags(&e, hipEventDisableTiming);
            h->seg_events.push_back(e);
        }
        rc = gpmpc_fit(h, hy.data(), 0, info);
        if (rc == GPMPC_OK) {
            hipFree(XT0); hipFree(Y0);
            ws_free(ws0);
            return GPMPC_OK;
        }
        const std::string keep = g_err;
        hipStreamSynchronize(h->stream);
        ws_free(h->ws);
        hipFree(XT1); hipFree(Y1);
        h->XT = XT0; h->Y = Y0; h->ws = ws0;
        h->N = N0; h->Np = Np0;
        h->hyper = hy;
        h->fitted = true;
        h->have_invK = invK0;
        h->have_beta = false;
        refresh_residual(h);
        g_err = keep;
        return rc;
    }
    const Ctx cx = h->cx();
    const long ld = Np1, sM = ws1.mat(), sW = ws1.wstride();
    for (int a = 0; a < Ny; ++a) {                          // unchanged rows < R0 of L and L^-1
        HIPCHK(hipMemcpy2DAsync(ws1.L + a * sM, ld * sizeof(double), h->ws.L + (size_t)a * Np0 * Np0, Np0 * sizeof(double),
                                R0 * sizeof(double), R0, hipMemcpyDeviceToDevice, cx.stream));
        HIPCHK(hipMemcpy2DAsync(ws1.Inv + a * sM, ld * sizeof(double), h->ws.Inv + (size_t)a * Np0 * Np0,
                                Np0 * sizeof(double), R0 * sizeof(double), R0, hipMemcpyDeviceToDevice, cx.stream));
    }
    HIPCHK(hipMemsetAsync(ws1.info, 0, Ny * sizeof(int), cx.stream));
    launch_gram(cx.stream, dim3(Np1 / 64, m / 64, Ny), d, XT1, ws1.hyper, ws1.jitter, ws1.K, N1, Np1, R0 / 64);
    const long oS = (long)R0 * ld;                          // first strip row
    {
        GemmP p = gemm_base(cx);                            // L21 = K21 inv11^T
        p.A = ws1.K + oS; p.lda = ld; p.sA = sM; p.a_mc = 0;
        p.B = ws1.Inv; p.ldb = ld; p.sB = sM; p.b_nc = 0; p.kflags = KB_LE_N;
        p.C = ws1.L + oS; p.ldc = ld; p.sC = sM;
        p.M = m; p.N = R0; p.K = R0;
        launch_gemm(p, Ny, cx.stream);
        GemmP q = gemm_base(cx);                            // S = K22 - L21 L21^T (lower)
        q.A = ws1.L + oS; q.lda = ld; q.sA = sM; q.a_mc = 0;
        q.B = ws1.L + oS; q.ldb = ld; q.sB = sM; q.b_nc = 0;
        q.C = ws1.K + oS + R0; q.ldc = ld; q.sC = sM;
        q.M = m; q.N = m; q.K = R0; q.alpha = -1.0; q.beta = 1.0; q.lower = 1;
        launch_gemm(q, Ny, cx.stream);
    }
    factor_blocked(cx, ws1, true, R0 / 64);                 // L22 and its diagonal-block inverses
    trtri_range(cx, ws1, cx.stream, R0, m);                 // inv22
    {
        double* W = ws1.W + ws1.hw() * ws1.hw();
        GemmP t = gemm_base(cx);                            // W = L21 inv11
        t.A = ws1.L + oS; t.lda = ld; t.sA = sM; t.a_mc = 0;
        t.B = ws1.Inv; t.ldb = ld; t.sB = sM; t.b_nc = 1; t.kflags = KB_GE_N;
        t.C = W; t.ldc = R0; t.sC = sW;
        t.M = m; t.N = R0; t.K = R0;
        launch_gemm(t, Ny, cx.stream);
        GemmP u = gemm_base(cx);                            // inv21 = -inv22 W
        u.A = ws1.Inv + oS + R0; u.lda = ld; u.sA = sM; u.a_mc = 0; u.kflags = KA_LE_M;
        u.B = W; u.ldb = R0; u.sB = sW; u.b_nc = 1;
        u.C = ws1.Inv + oS; u.ldc = ld; u.sC = sM;
        u.M = m; u.N = R0; u.K = m; u.alpha = -1.0;
        launch_gemm(u, Ny, cx.stream);
    }
    std::vector<int> inf(Ny, 0);
    HIPCHK(hipMemcpyAsync(inf.data(), ws1.info, Ny * sizeof(int), hipMemcpyDeviceToHost, cx.stream));
    HIPCHK(hipStreamSynchronize(cx.stream));
    HIPCHK(hipGetLastError());
    bool bad = false;
    for (int a = 0; a < Ny; ++a) {
        if (info) info[a] = inf[a] ? -inf[a] : 0;
        bad |= inf[a] != 0;
    }
    if (bad) {                                              // leave the model as it was
        ws_free(ws1);
        hipFree(XT1); hipFree(Y1);
        return fail(GPMPC_ENOTPD, "the extended K is not positive definite with the stored hyper-parameters and jitter");
    }
    install();
    CHK(refresh_residual(h));
    solve_alpha(h->cx(), h->ws, h->y_model(), h->Np);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    return GPMPC_OK;
}

// copy [Ny][Np x Np] device matrices to/from the caller's dense [Ny][N x N]
static int export_mats(gpmpc_gp* h, const double* dsrc, double* dst) {
    const int N = h->N, Np = h->Np;
    std::vector<double> tmp((size_t)Np * Np);
    for (int a = 0; a < h->Ny; ++a) {
        HIPCHK(hipMemcpy(tmp.data(), dsrc + (size_t)a * Np * Np, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < N; ++i)
            std::memcpy(dst + ((size_t)a * N + i) * N, tmp.data() + (size_t)i * Np, N * sizeof(double));
    }
    return GPMPC_OK;
}

static int import_mats(gpmpc_gp* h, const double* src, double* ddst, bool identity_pad) {
    const int N = h->N, Np = h->Np;
    std::vector<double> tmp((size_t)Np * Np);
    for (int a = 0; a < h->Ny; ++a) {
        std::fill(tmp.begin(), tmp.end(), 0.0);
        for (int i = 0; i < N; ++i)
            std::memcpy(tmp.data() + (size_t)i * Np, src + ((size_t)a * N + i) * N, N * sizeof(double));
        if (identity_pad)
            for (int i = N; i < Np; ++i) tmp[(size_t)i * Np + i] = 1.0;
        HIPCHK(hipMemcpy(ddst + (size_t)a * Np * Np, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return GPMPC_OK;
}

extern "C" int gpmpc_get_factors(gpmpc_gp* h, double* hyper, double* chol, double* alpha, double* invK) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    if (!h->fitted) return fail(GPMPC_ENOTFIT, "model has no factors (call gpmpc_fit or gpmpc_set_factors)");
    HIPCHK(hipSetDevice(h->device));
    alpha_ready(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    if (hyper) std::memcpy(hyper, h->hyper.data(), h->hyper.size() * sizeof(double));
    if (chol) CHK(export_mats(h, h->ws.L, chol));
    if (alpha) {
        std::vector<double> tmp((size_t)h->Ny * h->Np);
        HIPCHK(hipMemcpy(tmp.data(), h->ws.alpha, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int a = 0; a < h->Ny; ++a) std::memcpy(alpha + (size_t)a * h->N, tmp.data() + (size_t)a * h->Np, h->N * sizeof(double));
    }
    if (invK) {
        if (!h->have_invK) {
            CHK(compute_invK(h->cx(), h->ws));
            HIPCHK(hipStreamSynchronize(h->stream));
            h->have_invK = true;
        }
        CHK(export_mats(h, h->ws.InvK, invK));
    }
    return GPMPC_OK;
}

extern "C" int gpmpc_set_factors(gpmpc_gp* h, const double* hyper, const double* chol, const double* alpha,
                                 const double* invK) {
    if (!h || !hyper || !chol) return fail(GPMPC_EINVAL, "hyper and chol are required");
    HIPCHK(hipSetDevice(h->device));
    alpha_ready(h);
    h->tail.armed = false;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->fitted = false;
    h->have_invK = false;
    h->have_beta = false;
    h->hyper.assign(hyper, hyper + (size_t)h->Ny * h->nh());
    std::vector<double> kpart;
    CHK(upload_mean_and_residual(h, hyper, h->Ny, kpart, &h->mpar, h->Y, &h->Yc));
    HIPCHK(hipMemcpy(h->ws.hyper, kpart.data(), kpart.size() * sizeof(double), hipMemcpyHostToDevice));
    CHK(import_mats(h, chol, h->ws.L, true));
    factor_blocked(h->cx(), h->ws, false);  // L^-1 from the stored L
    if (alpha) {
        std::vector<double> tmp((size_t)h->Ny * h->Np, 0.0);
        for (int a = 0; a < h->Ny; ++a) std::memcpy(tmp.data() + (size_t)a * h->Np, alpha + (size_t)a * h->N, h->N * sizeof(double));
        HIPCHK(hipMemcpy(h->ws.alpha, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
        // the persistent variance product takes its mean from w = L^-1 y (api_predict.inl, fused mean): form it from the
        // imported factor even though the caller's alpha is kept as given
        hipLaunchKernelGGL(gemv_rows_kernel, dim3(h->Np / 4, h->ws.batch), dim3(256), 0, h->stream, h->ws.Inv, h->y_model(), h->ws.w,
                           h->Np, h->ws.mat(), (long)h->Np, (long)h->Np, 1);
    } else {
        solve_alpha(h->cx(), h->ws, h->y_model(), h->Np);
    }
    if (invK) {
        CHK(ws_need_invK(h->ws));
        CHK(import_mats(h, invK, h->ws.InvK, true));
        h->have_invK = true;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->fitted = true;
    return GPMPC_OK;
}

