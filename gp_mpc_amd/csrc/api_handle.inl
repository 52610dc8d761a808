// api_handle.inl -- part of gpmpc_api.hip (one translation unit; included in order, not compiled alone).
// Concern: model handle: create / destroy, setters, counters, profile read-out.
// ------------------------------------------------------------------------------------------------
// model handle
// ------------------------------------------------------------------------------------------------
struct gpmpc_gp {
    int device = 0, N = 0, Np = 0, d = 0, Ny = 0;
    hipStream_t own_stream = nullptr, stream = nullptr, side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_info = nullptr;       // "the factorisation's status words are on the host" (factor_with_jitter)
    int* pin = nullptr;                 // pinned host buffer for them
    double* roll_dev = nullptr;         // gpmpc_rollout: device staging [inputs | trajectories | scratch] (grow-only) ...
    double* roll_pin = nullptr;         // ... and its pinned mirror
    size_t roll_cap = 0;
    double* rollm_dev = nullptr;        // gpmpc_rollout_multi: the same for M trajectories in lock-step (grow-only)
    double* rollm_pin = nullptr;
    size_t rollm_cap = 0;
    struct RollGraph { std::vector<long> key; hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; };
    std::vector<RollGraph> roll_graphs; // captured T-step loops (launch-bound at small N), keyed by everything the launches depend on
    std::vector<long> roll_warm;        // key of the last plain run: a loop is captured only after it ran once uncaptured
    double* io_dev = nullptr;           // host-pointer mode, small calls: one device block [inputs | outputs] ...
    double* io_pin = nullptr;           // ... and its pinned host mirror: ONE copy each way instead of one per array
    size_t pin_ints = 0;
    hipStream_t aux_stream = nullptr, bulk_stream = nullptr;
    std::vector<hipEvent_t> seg_events;
    int chain_mode = 1;      // 0: single queue; 1-3: chained factorisation (gpmpc_create)
    // A hand-off time-out of the persistent kernels (GPU shared with work that keeps CUs from the workgroups that have
    // to be co-resident) repeats THIS factorisation on the single-queue path; the next call tries the chained path
    // again.  Only after CHAIN_STRIKES consecutive time-outs the handle stays on the single-queue path, and even then
    // it re-arms after CHAIN_REARM fits, so a transient neighbour does not cost a factor of two for ever.
    static constexpr int CHAIN_STRIKES = 3, CHAIN_REARM = 64;
    int chain_strikes = 0, chain_parked = 0;
    long n_timeouts = 0, n_chained = 0, n_single = 0;   // gpmpc_get_counter
    long n_var_persist = 0;                              // variance products through the persistent static-schedule kernel
    long n_fused = 0;                                    // gpmpc_fit_predict_mean_var calls that took the fused route
    long n_behind_tail = 0;                              // predictions that started next to a fit's tail (predict_behind_tail)
    long train_iters = 0, train_evals = 0;              // of the last gpmpc_train_multistart (this rank's restarts)
    double train_flop = 0.0;                            // ... and its algorithmic matrix flops: N^3/3 per Cholesky, per L^-1, per K^-1 lower triangle
    int nll_last_a = -1;                                 // the training workspace holds the factors of this output ...
    std::vector<double> nll_last_row;                    // ... at these hyper-parameters (gpmpc_nll; nll_grad_last reuses them)
#ifdef GPMPC_EMULATED
    int spin_limit = 1 << 30;   // the emulator's polls are scheduler passes, not time
#else
    int spin_limit = 40000;     // ~30 ms of polling with s_sleep before a waiter gives up: a thousand step times of the
                                // chain, and short enough for a control loop to survive the repeat on the other path
#endif
    int ptr_mode = GPMPC_PTR_HOST;
    int crow_mode = 0;
    bool fitted = false, have_invK = false;
    double *XT = nullptr, *Y = nullptr;  // [d][Np], [Ny][Np]
    Workspace ws;                        // model factors, batch = Ny
    Workspace tws;                       // training workspace, batch = 1 (lazy)
    Workspace bws;                       // lock-step restart search: batch = up to TRAIN_BATCH_CAP points of one output (lazy)
    double *bYc = nullptr, *bmpar = nullptr, *bgradPartial = nullptr, *bgradOut = nullptr;
    int* bzmap = nullptr;                                // slots of a subset of the batch (gradients of retained points)
    int bws_mem_cap = 0;                                 // > 0: the batch workspace was cut to this many points by the device's free memory
    struct LockRet { int pos = -1; std::vector<double> theta; };
    int lock_inv_panels = 0;                             // what the last value batch left in Inv (Workspace::inv_panels)
    std::vector<LockRet> lock_ret;                       // per restart of the lock-step search: where its last value-only point's factors are
    double* gradPartial = nullptr;
    double* gradOut = nullptr;
    std::vector<double> hyper;           // host copy [Ny][nh()]: [ell.., sf, sn, mean parameters]
    // prior mean function (gp_functions.py:25-69): kind GPMPC_MEAN_*, its parameters per output on the device,
    // and the residual targets y - m(X) that alpha and the NLL are formed from
    bool have_prior = false;             // Gaussian hyper-priors of calc_NLL (optimize.py:82-93)
    double prior[6] = {0, 1, 0, 1, 0, 1};  // ell_mean, ell_std, sf_mean, sf_std, sn_mean, sn_std
    int mean_kind = 0;
    bool mean_add = false;               // add m(z) to the predicted mean (build_gp's meanFunc argument)
    double* mpar = nullptr;              // [Ny][MPW]
    double* Yc = nullptr;                // [Ny][Np]
    double *tmpar = nullptr, *tYc = nullptr;   // the same for the single-output training workspace
    int nh() const { return d + 2 + mean_param_count(mean_kind, d); }
    const double* y_model() const { return mean_kind ? Yc : Y; }
    // predict scratch
    int Bcap = 0;
    double *Z = nullptr, *Sigma = nullptr, *KsT = nullptr, *part = nullptr, *meanT = nullptr;
    double* partm = nullptr;                     // the mean's partial sums of the persistent variance product (same shape as part)
    double *mean = nullptr, *var = nullptr, *J = nullptr, *cov = nullptr;
    double* em = nullptr;  // exact-moment / legacy scratch
    long emBytes = 0;
    double* ems = nullptr;   // scratch of gpmpc_predict_em_sens (grow-only)
    long emsBytes = 0;
    double* beta = nullptr;  // K^-1 y, [Ny][Np]
    double* UT = nullptr;    // K^-1 ks per test point (legacy methods, sensitivities)
    double* VT = nullptr;    // L^-1 ks per test point (sensitivities: K^-1 ks = L^-T (L^-1 ks) without K^-1)
    double *sensH = nullptr, *sensV = nullptr;   // staging of gpmpc_predict_sens outputs in host-pointer mode
    double* ccpart = nullptr;                    // chunk partials of the small-batch cross-covariance kernel
    bool have_beta = false;
    Prof prof;
    TailState tail;
    Ctx cx() {
        return Ctx{stream, crow_mode, side_stream, ev_fork, ev_join, chain_mode >= 2 ? aux_stream : nullptr,
                   seg_events.data(), (int)seg_events.size(), chain_mode >= 3 ? g_cu_count[device] : 0, &prof,
                   chain_mode >= 2 ? bulk_stream : nullptr, chain_mode >= 3 ? &tail : nullptr};
    }
};

struct PhaseTimer : ProfScope {
    PhaseTimer(gpmpc_gp* h, int ph) : ProfScope(&h->prof, h->stream, ph) {}
};

// alpha of the model workspace may still be in the making on the workers' queue (TailState): order the main queue behind it
static void alpha_ready(gpmpc_gp* h) {
    if (!h->tail.alpha_pending) return;
    hipStreamWaitEvent(h->stream, h->tail.ev_alpha, 0);
    h->tail.alpha_pending = false;
}

static int prof_collect(gpmpc_gp* h) {
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->side_stream) HIPCHK(hipStreamSynchronize(h->side_stream));   // (brackets of alpha / the cross-covariances next to a fit's tail)
    if (h->bulk_stream) HIPCHK(hipStreamSynchronize(h->bulk_stream));
    for (int ph = 0; ph < GPMPC_PH_COUNT; ++ph) {
        for (auto& pr : h->prof.ev[ph]) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, pr.first, pr.second);
            h->prof.total[ph] += ms;
            h->prof.count[ph] += 1;
            h->prof.pool.push_back(pr.first);
            h->prof.pool.push_back(pr.second);
        }
        h->prof.ev[ph].clear();
    }
    return GPMPC_OK;
}

// ---- prior mean function plumbing ------------------------------------------------------------------------------
// Splits host hyper rows [rows][nh] into the kernel part [rows][d+2] (what the SE-ARD kernels read) and uploads the
// mean parameters to `mpar_dev` ([rows][MPW]); then forms Yc = Y - m(X) for `rows` outputs starting at Y.
static int upload_mean_and_residual(gpmpc_gp* h, const double* hyper_rows, int rows, std::vector<double>& kernel_part,
                                    double** mpar_dev, const double* Y, double** Yc_dev) {
    const int d = h->d, nh = h->nh(), cnt = mean_param_count(h->mean_kind, d);
    kernel_part.resize((size_t)rows * (d + 2));
    for (int a = 0; a < rows; ++a) std::memcpy(&kernel_part[(size_t)a * (d + 2)], hyper_rows + (size_t)a * nh, (d + 2) * sizeof(double));
    if (!h->mean_kind) return GPMPC_OK;
    std::vector<double> mp((size_t)rows * MPW, 0.0);
    for (int a = 0; a < rows; ++a)
        for (int k = 0; k < cnt; ++k) {
            const double v = hyper_rows[(size_t)a * nh + d + 2 + k];
            if (!(v == v)) return fail(GPMPC_EINVAL, "mean-function parameter %d of row %d is NaN", k, a);
            mp[(size_t)a * MPW + k] = v;
        }
    if (!*mpar_dev) HIPCHK(hipMalloc(mpar_dev, (size_t)rows * MPW * sizeof(double)));
    if (!*Yc_dev) HIPCHK(hipMalloc(Yc_dev, (size_t)rows * h->Np * sizeof(double)));
    HIPCHK(hipStreamSynchronize(h->stream));     // `mp` is a stack-lifetime source: the copy below must not outlive it
    HIPCHK(hipMemcpy(*mpar_dev, mp.data(), mp.size() * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mean_resid_kernel, dim3((h->Np + 255) / 256, rows), dim3(256), 0, h->stream, h->XT, Y, *mpar_dev, *Yc_dev,
                       h->mean_kind, h->N, h->Np, d, (long)h->Np);
    return GPMPC_OK;
}

extern "C" {

int gpmpc_abi_version(void) { return GPMPC_ABI_VERSION; }
const char* gpmpc_last_error(void) { return g_err.c_str(); }

int gpmpc_device_count(int* count) {
    if (!count) return fail(GPMPC_EINVAL, "count is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return GPMPC_OK;
}

int gpmpc_device_name(int device, char* buf, int buflen) {
    if (!buf || buflen <= 0) return fail(GPMPC_EINVAL, "bad buffer");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return GPMPC_OK;
}

int gpmpc_mfma_selftest(int device, int* layout_out, double* tflops_out) {
    CHK(ensure_device(device));
    return mfma_selftest(device, layout_out, tflops_out);
}

int gpmpc_destroy(gpmpc_gp* h);
}  // extern "C"

// events for hand-overs between the queues of the factorisation: segments of the pipelined inverse, or two per
// super-panel of the two-level execution (>= 2 block columns each; two more each with the look-ahead, one for the blocked rows below)
static size_t seg_event_count(int Np) { return (size_t)std::max(3, std::max(Np / SEGR + 2, 3 * (Np / 64) + 8)); }

static int create_impl(gpmpc_gp* h, const double* X, const double* Y) {
    const int device = h->device, N = h->N, d = h->d, Ny = h->Ny;
    h->crow_mode = g_crow_mode[device];
    HIPCHK(hipStreamCreate(&h->own_stream));
    h->stream = h->own_stream;
    HIPCHK(hipStreamCreate(&h->side_stream));
    HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    {
        // (tuning aid) GPMPC_AUX_PRIORITY=low|high: priority of the inverse queue (default: normal)
        const char* ap = getenv("GPMPC_AUX_PRIORITY");
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        if (ap && (ap[0] == 'l' || ap[0] == 'h')) HIPCHK(hipStreamCreateWithPriority(&h->aux_stream, hipStreamDefault, ap[0] == 'l' ? lo : hi));
        else HIPCHK(hipStreamCreate(&h->aux_stream));
    }
    {
        int lo = 0, hi = 0;                                    // (numerically larger = lower priority)
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&h->bulk_stream, hipStreamDefault, lo));
    }
    // the persistent kernels ask for more than the default 64 KB of dynamic LDS (per device: set for every handle)
    HIPCHK(hipFuncSetAttribute((const void*)chol_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CHAIN_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)chol_worker_kernel<WORKER_MAXT, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               WORKER_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)chol_worker_kernel<WORKER_MAXT_COURIER, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               WORKER_LDS_BYTES));
    const size_t nseg = seg_event_count(round_up(N, 64));
    for (size_t i = 0; i < nseg; ++i) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->seg_events.push_back(e);
    }
    if (getenv("GPMPC_CHAIN_TRACE") && !g_chain_trace) {
        HIPCHK(hipMalloc(&g_chain_trace, (size_t)(1 << 20) * sizeof(long long)));
        HIPCHK(hipMemset(g_chain_trace, 0, (size_t)(1 << 20) * sizeof(long long)));
    }
    // 0: single queue; 1: chained Cholesky, bulk in GEMM launches; 2: + inverse pipelined behind the chain;
    // 3: + bulk in the persistent tile-owner kernel where the matrix fits its registers (else as 2)
    h->chain_mode = 3;
    if (const char* e = getenv("GPMPC_CHAIN")) h->chain_mode = atoi(e);
    if (const char* e = getenv("GPMPC_SPIN_LIMIT")) h->spin_limit = atoi(e);   // tests: force the hand-off time-out path
    const int Np = h->Np;
    std::vector<double> xt((size_t)d * Np, 0.0), yt((size_t)Ny * Np, 0.0);
    for (int i = 0; i < N; ++i) {
        for (int k = 0; k < d; ++k) xt[(size_t)k * Np + i] = X[(size_t)i * d + k];
        for (int a = 0; a < Ny; ++a) yt[(size_t)a * Np + i] = Y[(size_t)i * Ny + a];
    }
    HIPCHK(hipMalloc(&h->XT, xt.size() * sizeof(double)));
    HIPCHK(hipMalloc(&h->Y, yt.size() * sizeof(double)));
    HIPCHK(hipMemcpy(h->XT, xt.data(), xt.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->Y, yt.data(), yt.size() * sizeof(double), hipMemcpyHostToDevice));
    CHK(ws_alloc(h->ws, Ny, Np, d));
    h->hyper.assign((size_t)Ny * (d + 2), 0.0);
    return GPMPC_OK;
}

extern "C" {

int gpmpc_create(int device, int N, int d, int Ny, const double* X, const double* Y, gpmpc_gp** out) {
    if (!out) return fail(GPMPC_EINVAL, "out is NULL");
    *out = nullptr;
    if (N <= 0 || d <= 0 || Ny <= 0 || !X || !Y) return fail(GPMPC_EINVAL, "bad N/d/Ny or NULL data");
    if (d > DMAX) return fail(GPMPC_EINVAL, "input dimension d=%d exceeds the built-in maximum %d", d, DMAX);
    CHK(ensure_device(device));
    gpmpc_gp* h = new gpmpc_gp();
    {
        std::lock_guard<std::mutex> lk(g_block_mutex);
        ++g_live_handles;
    }
    h->device = device; h->N = N; h->d = d; h->Ny = Ny; h->Np = round_up(N, 64);
    const int rc = create_impl(h, X, Y);
    if (rc != GPMPC_OK) {                       // every early exit releases what was created so far
        const std::string keep = g_err;
        gpmpc_destroy(h);
        g_err = keep;
        return rc;
    }
    *out = h;
    return GPMPC_OK;
}

static void drop_roll_graphs(gpmpc_gp* h) {
#ifndef GPMPC_EMULATED
    for (auto& g : h->roll_graphs) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
    }
#endif
    h->roll_graphs.clear();
    h->roll_warm.clear();
}

int gpmpc_destroy(gpmpc_gp* h) {
    if (!h) return GPMPC_OK;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->side_stream) hipStreamSynchronize(h->side_stream);
    if (h->bulk_stream) hipStreamSynchronize(h->bulk_stream);
    ws_free(h->ws);
    ws_free(h->tws);
    ws_free(h->bws);
    hipFree(h->bYc); hipFree(h->bmpar); hipFree(h->bgradPartial); hipFree(h->bgradOut); hipFree(h->bzmap);
    hipFree(h->XT); hipFree(h->Y); hipFree(h->gradPartial); hipFree(h->gradOut);
    hipFree(h->mpar); hipFree(h->Yc); hipFree(h->tmpar); hipFree(h->tYc);
    hipFree(h->Z); hipFree(h->Sigma); hipFree(h->KsT); hipFree(h->part); hipFree(h->partm); hipFree(h->meanT);
    hipFree(h->mean); hipFree(h->var); hipFree(h->J); hipFree(h->cov); hipFree(h->em); hipFree(h->ems);
    hipFree(h->beta); hipFree(h->UT); hipFree(h->VT); hipFree(h->sensH); hipFree(h->sensV); hipFree(h->ccpart);
    for (int ph = 0; ph < GPMPC_PH_COUNT; ++ph)
        for (auto& pr : h->prof.ev[ph]) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto e : h->prof.pool) hipEventDestroy(e);
    if (h->ev_info) hipEventDestroy(h->ev_info);
    for (hipEvent_t e : {h->tail.ev_chain, h->tail.ev_tail, h->tail.ev_alpha, h->tail.ev_ks, h->tail.ev_mean, h->tail.ev_w})
        if (e) hipEventDestroy(e);
    if (h->pin) hipHostFree(h->pin);
    if (h->io_pin) hipHostFree(h->io_pin);
    hipFree(h->io_dev);
    drop_roll_graphs(h);
    if (h->roll_pin) hipHostFree(h->roll_pin);
    hipFree(h->roll_dev);
    if (h->rollm_pin) hipHostFree(h->rollm_pin);
    hipFree(h->rollm_dev);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    for (auto e : h->seg_events) hipEventDestroy(e);
    if (h->aux_stream) hipStreamDestroy(h->aux_stream);
    if (h->bulk_stream) hipStreamDestroy(h->bulk_stream);
    if (h->side_stream) hipStreamDestroy(h->side_stream);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
    bool last;
    {
        std::lock_guard<std::mutex> lk(g_block_mutex);
        last = --g_live_handles == 0;
    }
    if (last) block_list_release();
    return GPMPC_OK;
}

int gpmpc_get_size(const gpmpc_gp* h, int* N, int* d, int* Ny) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    if (N) *N = h->N;
    if (d) *d = h->d;
    if (Ny) *Ny = h->Ny;
    return GPMPC_OK;
}

int gpmpc_set_mean_func(gpmpc_gp* h, int kind, int add_to_prediction) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    if (kind < GPMPC_MEAN_ZERO || kind > GPMPC_MEAN_POLYNOMIAL) return fail(GPMPC_EINVAL, "No mean function with code %d", kind);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->mean_kind = kind;
    h->mean_add = add_to_prediction != 0;
    h->nll_last_a = -1;                                  // (the training workspace's factors belong to the old objective)
    h->hyper.assign((size_t)h->Ny * h->nh(), 0.0);     // rows change width: the model has to be fitted / loaded again
    h->fitted = false;
    h->have_invK = false;
    h->have_beta = false;
    return GPMPC_OK;
}

int gpmpc_set_hyper_prior(gpmpc_gp* h, const double* prior6) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    h->have_prior = prior6 != nullptr;
    h->nll_last_a = -1;                                  // (nll_grad_last adds the prior's gradient of the point gpmpc_nll saw)
    if (prior6) {
        for (int k = 0; k < 6; ++k) {
            if (!(prior6[k] == prior6[k]) || ((k & 1) && !(prior6[k] > 0.0)))
                return fail(GPMPC_EINVAL, "prior[%d] = %g: means must be numbers, standard deviations positive", k, prior6[k]);
            h->prior[k] = prior6[k];
        }
    }
    return GPMPC_OK;
}

int gpmpc_hyper_width(const gpmpc_gp* h, int* width) {
    if (!h || !width) return fail(GPMPC_EINVAL, "NULL handle/width");
    *width = h->nh();
    return GPMPC_OK;
}

int gpmpc_set_pointer_mode(gpmpc_gp* h, int mode) {
    if (!h || (mode != GPMPC_PTR_HOST && mode != GPMPC_PTR_DEVICE)) return fail(GPMPC_EINVAL, "bad pointer mode");
    h->ptr_mode = mode;
    return GPMPC_OK;
}

int gpmpc_set_stream(gpmpc_gp* h, void* s) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    HIPCHK(hipStreamSynchronize(h->stream));
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return GPMPC_OK;
}

int gpmpc_synchronize(gpmpc_gp* h) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return GPMPC_OK;
}

int gpmpc_get_counter(gpmpc_gp* h, const char* name, long* value) {
    if (!h || !name || !value) return fail(GPMPC_EINVAL, "NULL argument");
    if (std::strcmp(name, "handoff_timeouts") == 0) *value = h->n_timeouts;
    else if (std::strcmp(name, "chained_factorisations") == 0) *value = h->n_chained;
    else if (std::strcmp(name, "single_queue_factorisations") == 0) *value = h->n_single;
    else if (std::strcmp(name, "predictions_behind_tail") == 0) *value = h->n_behind_tail;
    else if (std::strcmp(name, "fused_fit_predicts") == 0) *value = h->n_fused;
    else if (std::strcmp(name, "train_gflop") == 0) *value = (long)(h->train_flop * 1e-9 + 0.5);
    else if (std::strcmp(name, "persistent_variance_products") == 0) *value = h->n_var_persist;
    else if (std::strcmp(name, "train_iterations") == 0) *value = h->train_iters;
    else if (std::strcmp(name, "train_evaluations") == 0) *value = h->train_evals;
    else if (std::strcmp(name, "workspace_blocks_reused") == 0 || std::strcmp(name, "workspace_blocks_fresh") == 0) {
        std::lock_guard<std::mutex> lk(g_block_mutex);
        *value = std::strcmp(name, "workspace_blocks_reused") == 0 ? g_block_reuses : g_block_fresh;
    }
    else return fail(GPMPC_EINVAL, "unknown counter '%s'", name);
    return GPMPC_OK;
}

int gpmpc_profile_enable(gpmpc_gp* h, int enable) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    h->prof.on = enable != 0;                  // (any nonzero value: on; which phases: gpmpc_profile_set_mask)
    return GPMPC_OK;
}

int gpmpc_profile_set_mask(gpmpc_gp* h, unsigned mask) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    h->prof.mask = mask ? mask : ~0u;
    return GPMPC_OK;
}

int gpmpc_profile_read(gpmpc_gp* h, int phase, double* total_ms, long* launches, int reset) {
    if (!h || phase < 0 || phase >= GPMPC_PH_COUNT) return fail(GPMPC_EINVAL, "bad phase");
    CHK(prof_collect(h));
    if (total_ms) *total_ms = h->prof.total[phase];
    if (launches) *launches = h->prof.count[phase];
    if (reset) { h->prof.total[phase] = 0.0; h->prof.count[phase] = 0; }
    return GPMPC_OK;
}

}  // extern "C"

