// api_train.inl -- part of gpmpc_api.hip (one translation unit; included in order, not compiled alone).
// Concern: NLL (+ gradient), multistart training, RCCL binding of the restart shard.
// ------------------------------------------------------------------------------------------------
// a7 NLL (+ analytic gradient) on the separate single-output training workspace
// ------------------------------------------------------------------------------------------------
// gpmpc_set_tuning("fail_nll_after", n): the n-th NLL evaluation of the process from now on reports a device failure
// (GPMPC_EHIP) without touching the device -- how the tests reach the failure paths of the restart shard (0 = off).
// Only in a process started with GPMPC_TESTING=1 (the knob is refused otherwise): production code cannot arm it.
static std::atomic<int> g_fail_nll_after{0};

// K^-1 (lower triangle) from the L^-1 in `ws`, then the gradient reductions into h->gradOut: what gpmpc_nll adds for a
// gradient, enqueued on the handle's stream
static void enqueue_nll_grad(gpmpc_gp* h, Workspace& ws, const Ctx& cx, int nmean) {
    const int d = h->d, Np = h->Np;
    {
        PhaseTimer t(h, GPMPC_PH_INVK);
        (void)invk_lower(cx, ws, 1);   // the lower triangle of K^-1 = L^-T L^-1 is all the gradient pass reads
    }
    PhaseTimer t(h, GPMPC_PH_NLL);
    hipLaunchKernelGGL(nll_grad_kernel, dim3(Np / 64, Np / 64), dim3(256), 0, cx.stream, h->XT, ws.hyper, ws.InvK,
                       ws.alpha, h->gradPartial, h->N, Np, d);
    hipLaunchKernelGGL(nll_grad_finish_kernel, dim3(1), dim3(256), 0, cx.stream, h->gradPartial, ws.hyper,
                       h->gradOut, Np, d);
    if (nmean)
        hipLaunchKernelGGL(mean_grad_kernel, dim3(1), dim3(256), 0, cx.stream, h->XT, ws.alpha, h->gradOut + d + 2,
                           h->mean_kind, h->N, Np, d);
}

// calc_NLL optimize.py:77-97, literally: `return NLL(...) + log_prior` with log_prior the sum of the Gaussian
// log-densities prior_gauss(theta, mu, s^2) = -(theta - mu)^2 / (2 s^2) - 1/2 log(2 pi s^2) of every ell_i and of
// sf^2 and sn^2 (the SQUARED hyper-parameters, :90-91).  (The log-prior is ADDED to the negative log-likelihood
// there, not subtracted; the reference never enables it, prior = None :157.)  Either output may be NULL.
static void add_log_prior(const gpmpc_gp* h, const double* hyper_row, double* nll, double* grad) {
    if (!h->have_prior) return;
    const int d = h->d;
    const double two_pi = 6.283185307179586476925286766559;
    auto lg = [&](double th, double mu, double sd) { return -(th - mu) * (th - mu) / (2.0 * sd * sd) - 0.5 * std::log(two_pi * sd * sd); };
    auto dlg = [&](double th, double mu, double sd) { return -(th - mu) / (sd * sd); };
    double lp = 0.0;
    for (int k = 0; k < d; ++k) {
        lp += lg(hyper_row[k], h->prior[0], h->prior[1]);
        if (grad) grad[k] += dlg(hyper_row[k], h->prior[0], h->prior[1]);
    }
    const double sf = hyper_row[d], sn = hyper_row[d + 1];
    lp += lg(sf * sf, h->prior[2], h->prior[3]) + lg(sn * sn, h->prior[4], h->prior[5]);
    if (grad) {
        grad[d] += dlg(sf * sf, h->prior[2], h->prior[3]) * 2.0 * sf;
        grad[d + 1] += dlg(sn * sn, h->prior[4], h->prior[5]) * 2.0 * sn;
    }
    if (nll) *nll += lp;
}

// The gradient at the point gpmpc_nll evaluated LAST (value only) on this handle: the factors are still in the training
// workspace, so this is the K^-1 product and the reductions, not a second factorisation.  GPMPC_EINVAL if the workspace
// holds something else.  (A line search evaluates values until a step is accepted and asks for one gradient then: a
// rejected trial point costs 1.7 instead of 2.5 ms at N = 4096; same numbers as gpmpc_nll with a gradient, same kernels.)
static int nll_grad_last(gpmpc_gp* h, int a, const double* hyper_row, double* grad) {
    const int d = h->d, nh = h->nh();
    if (!h->tws.K || h->nll_last_a != a || (int)h->nll_last_row.size() != nh ||
        std::memcmp(h->nll_last_row.data(), hyper_row, nh * sizeof(double)) != 0)
        return fail(GPMPC_EINVAL, "the training workspace does not hold this point");
    HIPCHK(hipSetDevice(h->device));
    Workspace& ws = h->tws;
    CHK(ws_need_invK(ws));
    const int nmean = mean_param_count(h->mean_kind, d);
    enqueue_nll_grad(h, ws, h->cx(), nmean);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(grad, h->gradOut, (d + 2 + nmean) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    add_log_prior(h, hyper_row, nullptr, grad);
    return GPMPC_OK;
}

extern "C" int gpmpc_nll(gpmpc_gp* h, int a, const double* hyper_row, double* nll, double* grad, int* jitter_out) {
    if (!h || !hyper_row || !nll || a < 0 || a >= h->Ny) return fail(GPMPC_EINVAL, "bad arguments");
    if (g_fail_nll_after.load(std::memory_order_relaxed) > 0 && g_fail_nll_after.fetch_sub(1) == 1)
        return fail(GPMPC_EHIP, "injected device failure (fail_nll_after)");
    HIPCHK(hipSetDevice(h->device));
    const int d = h->d, Np = h->Np;
    for (int k = 0; k < d + 1; ++k)
        if (!(hyper_row[k] == hyper_row[k]) || hyper_row[k] == 0.0)
            return fail(GPMPC_EINVAL, "hyper_row[%d] = %g is not a usable SE-ARD parameter", k, hyper_row[k]);
    if (!h->tws.K) {
        CHK(ws_alloc(h->tws, 1, Np, d));
        HIPCHK(hipMalloc(&h->gradPartial, (size_t)(Np / 64) * (Np / 64) * (DMAX + 2) * sizeof(double)));
        HIPCHK(hipMalloc(&h->gradOut, (DMAX + 2 + MPW) * sizeof(double)));
    }
    Workspace& ws = h->tws;
    int info = 0;
    const Ctx cx = h->cx();
    h->nll_last_a = -1;                       // (set again when this evaluation has succeeded)
    if (grad) CHK(ws_need_invK(ws));
    // prior mean: the objective is evaluated on y - m(X) (calc_NLL optimize.py:43,75,96)
    std::vector<double> kpart;
    CHK(upload_mean_and_residual(h, hyper_row, 1, kpart, &h->tmpar, h->Y + (size_t)a * Np, &h->tYc));
    const double* ytrain = h->mean_kind ? h->tYc : h->Y + (size_t)a * Np;
    const int nmean = mean_param_count(h->mean_kind, d);
    // everything that follows the factorisation is enqueued before the host waits for `info` (factor_with_jitter)
    CHK(factor_with_jitter(h, ws, hyper_row, &info, [&]() {
        {
            PhaseTimer t(h, GPMPC_PH_SOLVE);
            solve_alpha(cx, ws, ytrain, Np);
        }
        {
            PhaseTimer t(h, GPMPC_PH_NLL);
            hipLaunchKernelGGL(nll_reduce_kernel, dim3(1), dim3(256), 0, cx.stream, ws.L, ws.w, ws.nll, h->N, Np);
        }
        if (grad) enqueue_nll_grad(h, ws, cx, nmean);
    }));
    if (jitter_out) *jitter_out = info;
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(nll, ws.nll, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (grad) HIPCHK(hipMemcpyAsync(grad, h->gradOut, (d + 2 + nmean) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    add_log_prior(h, hyper_row, nll, grad);
    h->nll_last_a = a;
    h->nll_last_row.assign(hyper_row, hyper_row + h->nh());
    return GPMPC_OK;
}

// ------------------------------------------------------------------------------------------------
// a8 multistart training behind the C ABI (train_gp_numpy optimize.py:359-503 / train_gp :100-294)
// ------------------------------------------------------------------------------------------------
extern "C" int gpmpc_rccl_unique_id(char* id128) {
    if (!id128) return fail(GPMPC_EINVAL, "NULL id buffer");
    RcclApi& R = rccl_api();
    if (!R.ok()) return fail(GPMPC_EHIP, "librccl.so could not be loaded: %s", R.load_error.c_str());
    RcclId id;
    const int rc = R.GetUniqueId(&id);
    if (rc != 0) return fail(GPMPC_EHIP, "ncclGetUniqueId failed: %s", R.GetErrorString ? R.GetErrorString(rc) : "?");
    std::memcpy(id128, id.internal, 128);
    return GPMPC_OK;
}

extern "C" int gpmpc_rccl_comm_create(int device, int world, int rank, const char* id128, void** comm_out) {
    if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) return fail(GPMPC_EINVAL, "bad arguments");
    *comm_out = nullptr;
    CHK(ensure_device(device));
    RcclApi& R = rccl_api();
    if (!R.ok()) return fail(GPMPC_EHIP, "librccl.so could not be loaded: %s", R.load_error.c_str());
    RcclId id;
    std::memcpy(id.internal, id128, 128);
    HIPCHK(hipDeviceSynchronize());
    (void)hipGetLastError();      // RCCL treats a stale "last error" of this thread (e.g. hipErrorNotReady of an event query) as its own
    const int rc = R.CommInitRank(comm_out, world, id, rank);
    if (rc != 0) return fail(GPMPC_EHIP, "ncclCommInitRank failed: %s", R.GetErrorString ? R.GetErrorString(rc) : "?");
    return GPMPC_OK;
}

extern "C" int gpmpc_rccl_comm_destroy(void* comm) {
    if (!comm) return GPMPC_OK;
    RcclApi& R = rccl_api();
    if (!R.ok()) return fail(GPMPC_EHIP, "librccl.so could not be loaded");
    return R.CommDestroy(comm) == 0 ? GPMPC_OK : fail(GPMPC_EHIP, "ncclCommDestroy failed");
}

extern "C" int gpmpc_rccl_comm_count(void* comm, int* count) {
    if (!comm || !count) return fail(GPMPC_EINVAL, "NULL communicator or output");
    RcclApi& R = rccl_api();
    if (!R.ok() || !R.CommCount) return fail(GPMPC_EHIP, "librccl.so could not be loaded: %s", R.load_error.c_str());
    const int rc = R.CommCount(comm, count);
    return rc == 0 ? GPMPC_OK : fail(GPMPC_EHIP, "ncclCommCount failed: %s", R.GetErrorString ? R.GetErrorString(rc) : "?");
}

// ------------------------------------------------------------------------------------------------
// Lock-step restart search (r04): many hyper-parameter points of ONE output evaluated as one batch
// ------------------------------------------------------------------------------------------------
// A restart's search evaluates the NLL at one point at a time, and one 4096^2 factorisation is latency-bound (64
// sequential leaves: 0.35 of the fp64 MFMA peak for value + gradient, r03).  The restarts of a rank are independent, so
// they advance in lock-step: every live restart runs its unchanged projected L-BFGS in a thread of its own, an
// evaluation request parks the thread, and when all live restarts are parked the points they ask for go through the
// device as ONE batch -- K build, two-level factorisation (factor_twolevel: the execution that gives 0.6 of peak on
// batches), alpha, log det, and for the points that want it the K^-1 product and the gradient pass, all with a batch
// dimension.  A point's result does not depend on what else is in its batch (no tile-owner workers here: the execution is
// chosen without looking at the batch size; every batched kernel treats its matrices independently), so the search of a
// restart is the same whether it runs alone or next to 63 others -- bitwise: the restart shard stays world-size invariant.
// Line-search trials ask for the value only (K build, factorisation with L^-1, alpha, log det); the batch leaves every
// point's factors in its slot of the batch workspace, and the restart whose trial is accepted asks for the gradient of
// that point in the next round: K^-1 and the gradient pass on the retained L^-1 of the accepted points only (`zmap`: a
// subset of the slots), before the round's new points overwrite the slots.  130 of the 450 evaluations of the 64-restart
// C4 run are rejected trials: they cost 0.64 of a value + gradient evaluation.  (When the restarts of a rank do not fit one
// batch the slots would be overwritten in between: every trial is evaluated with its gradient then, as before.)
constexpr int TRAIN_BATCH_CAP_MAX = 64;
static int g_train_batch_cap = 0;            // gpmpc_set_tuning("train_batch_cap", n): 0 = automatic (memory), 1 = one point at a time
static constexpr int BGS = DMAX + 2 + MPW;   // stride of a point's gradient in bgradOut

struct NllReq {
    const double* theta = nullptr;           // [nh]
    int id = 0;                              // the restart (its slot in the pool)
    bool grad_of_last = false;               // no evaluation: the gradient at the point this restart evaluated last (value only)
    bool want_grad = false;
    double f = std::numeric_limits<double>::infinity();
    double* g = nullptr;                     // [nh], want_grad
    int rc = GPMPC_OK;                       // GPMPC_OK, GPMPC_ENOTPD (unusable point), GPMPC_EINVAL, or a device failure
};

static int ensure_batch_ws(gpmpc_gp* h, int want) {
    const int Np = h->Np, d = h->d;
    // per point: K, L, L^-1, K^-1 and the inverse's scratch
    Workspace probe;
    probe.Np = Np;
    const double per_point = (4.0 * Np * Np + (double)probe.wstride()) * 8.0;
    int cap = (int)std::min<double>(TRAIN_BATCH_CAP_MAX, std::max(1.0, 64.0e9 / per_point));
    if (g_train_batch_cap > 0) cap = std::min(cap, g_train_batch_cap);
    want = std::max(1, std::min(want, cap));
    // (a request that was cut to what the device's memory allowed comes back with every batch of the same size: it is met by
    //  the workspace that cut produced -- without this the early-out below compared against the uncut request and every call
    //  released and re-allocated the whole workspace to end at the same size)
    if (h->bws.K && h->bws_mem_cap > 0 && want > h->bws_mem_cap && h->bws.batch >= h->bws_mem_cap) return GPMPC_OK;
    if (h->bws.K && h->bws.batch >= want) return GPMPC_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    ws_free(h->bws);
    hipFree(h->bYc); hipFree(h->bmpar); hipFree(h->bgradPartial); hipFree(h->bgradOut); hipFree(h->bzmap);
    h->bYc = h->bmpar = h->bgradPartial = h->bgradOut = nullptr;
    h->bzmap = nullptr;
    const int want_before_mem = want;
    {
        // what the device has free now (the old batch workspace is released), less 10 % and 1 GB of headroom: a smaller or
        // busier device gets a smaller batch instead of GPMPC_ENOMEM
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) {
            const double usable = 0.9 * (double)(free_b + block_list_idle_bytes(h->device)) - 1.0e9;
            want = std::max(1, std::min(want, (int)std::max(1.0, usable / per_point)));
        }
    }
    int arc = ws_alloc(h->bws, want, Np, d);
    while (arc != GPMPC_OK && want > 1) {        // still too large (fragmentation): halve until it fits
        ws_free(h->bws);
        want = (want + 1) / 2;
        arc = ws_alloc(h->bws, want, Np, d);
    }
    CHK(arc);
    h->bws_mem_cap = want < want_before_mem ? want : 0;      // > 0: this device's memory holds no more than that many points
    HIPCHK(hipMalloc(&h->bzmap, (size_t)want * sizeof(int)));
    CHK(ws_need_invK(h->bws));
    HIPCHK(hipMalloc(&h->bgradPartial, (size_t)want * (Np / 64) * (Np / 64) * (DMAX + 2) * sizeof(double)));
    HIPCHK(hipMalloc(&h->bgradOut, (size_t)want * BGS * sizeof(double)));
    HIPCHK(hipMalloc(&h->bYc, (size_t)want * Np * sizeof(double)));
    HIPCHK(hipMalloc(&h->bmpar, (size_t)want * MPW * sizeof(double)));
    const size_t need = seg_event_count(Np);
    while (h->seg_events.size() < need) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->seg_events.push_back(e);
    }
    return GPMPC_OK;
}

// n <= h->bws.batch points of output a, all with or all without the gradient; jit: the jitter added to every K of this
// call (0, or 1e-8 for the repeat of the points whose first factorisation failed: optimize.py:345-350 per point).
// value_only (a line-search trial whose factors are kept for a possible gradient request): the factorisation may leave L^-1
// unformed -- h->lock_inv_panels says what it left -- and alpha is not formed either.
static int nll_batch_core(gpmpc_gp* h, int a, int n, NllReq* const* req, bool want_grad, double jit, std::vector<int>& failed,
                          bool value_only = false) {
    const int d = h->d, Np = h->Np, nh = h->nh(), nmean = mean_param_count(h->mean_kind, d);
    Workspace ws = h->bws;                       // a view: the first n matrices
    ws.batch = n;
    Ctx cx = h->cx();
    cx.no_workers = true;
    std::vector<double> kpart((size_t)n * (d + 2));
    for (int i = 0; i < n; ++i) std::memcpy(&kpart[(size_t)i * (d + 2)], req[i]->theta, (d + 2) * sizeof(double));
    const double* ytrain = h->Y + (size_t)a * Np;
    long sy = 0;                                 // every point shares the output's targets ...
    if (h->mean_kind) {                          // ... unless a prior mean is trained: y - m(X) per point (calc_NLL optimize.py:43,75,96)
        std::vector<double> mp((size_t)n * MPW, 0.0);
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < nmean; ++k) mp[(size_t)i * MPW + k] = req[i]->theta[d + 2 + k];
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(h->bmpar, mp.data(), mp.size() * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(mean_resid_kernel, dim3((Np + 255) / 256, n), dim3(256), 0, h->stream, h->XT, ytrain, h->bmpar, h->bYc,
                           h->mean_kind, h->N, Np, d, 0L);
        ytrain = h->bYc;
        sy = Np;
    }
    std::vector<int> info(n, 0);
    const int frc = factor_with_jitter(h, ws, kpart.data(), info.data(), [&]() {
        {
            // w = L^-1 y by blocked substitution (the same arithmetic whether or not this point's L^-1 exists), alpha only with it
            PhaseTimer t(h, GPMPC_PH_SOLVE);
            fwd_subst(cx, ws, twolevel_width(Np) > 1 ? twolevel_width(Np) : 8, ytrain, sy, ws.alpha);
            if (!ws.inv_panels) solve_alpha_from_w(cx, ws, n, nullptr);
        }
        {
            PhaseTimer t(h, GPMPC_PH_NLL);
            hipLaunchKernelGGL(nll_reduce_kernel, dim3(n), dim3(256), 0, cx.stream, ws.L, ws.w, ws.nll, h->N, Np);
        }
        if (want_grad) {
            {
                PhaseTimer t(h, GPMPC_PH_INVK);
                // the lower triangle of K^-1 = L^-T L^-1 is all the gradient pass reads; an element's bits do not depend on n
                (void)invk_lower(cx, ws, n);
            }
            PhaseTimer t(h, GPMPC_PH_NLL);
            hipLaunchKernelGGL(nll_grad_kernel, dim3(Np / 64, Np / 64, n), dim3(256), 0, cx.stream, h->XT, ws.hyper, ws.InvK,
                               ws.alpha, h->bgradPartial, h->N, Np, d);
            hipLaunchKernelGGL(nll_grad_finish_kernel, dim3(n), dim3(256), 0, cx.stream, h->bgradPartial, ws.hyper, h->bgradOut,
                               Np, d, BGS);
            if (nmean)
                hipLaunchKernelGGL(mean_grad_kernel, dim3(n), dim3(256), 0, cx.stream, h->XT, ws.alpha, h->bgradOut + d + 2,
                                   h->mean_kind, h->N, Np, d, BGS);
        }
    }, 1, jit, true, value_only && !want_grad);
    h->lock_inv_panels = ws.inv_panels;
    {   // algorithmic matrix flops of this batch (counter "train_gflop"): Cholesky, L^-1 unless left unformed, K^-1 with a gradient
        const double n3 = (double)h->N * h->N * h->N / 3.0;
        h->train_flop += n * n3 * (1.0 + (ws.inv_panels ? 0.0 : 1.0) + (want_grad ? 1.0 : 0.0));
    }
    if (frc != GPMPC_OK && frc != GPMPC_ENOTPD) return frc;
    HIPCHK(hipGetLastError());
    std::vector<double> fv(n), gv(want_grad ? (size_t)n * BGS : 0);
    HIPCHK(hipMemcpyAsync(fv.data(), ws.nll, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (want_grad) HIPCHK(hipMemcpyAsync(gv.data(), h->bgradOut, gv.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; ++i) {
        if (info[i] < 0) { failed.push_back(i); continue; }
        req[i]->rc = GPMPC_OK;
        req[i]->f = fv[i];
        if (want_grad) std::memcpy(req[i]->g, &gv[(size_t)i * BGS], nh * sizeof(double));
    }
    return GPMPC_OK;
}

// The gradients of points whose factors a value-only batch left in the batch workspace (slots pos[i]): K^-1 of those
// matrices and the gradient pass, nothing else.  Same kernels on the same factors as a value + gradient evaluation.
static int nll_grad_retained(gpmpc_gp* h, int a, const std::vector<NllReq*>& G, const std::vector<int>& pos) {
    (void)a;
    const int d = h->d, Np = h->Np, nh = h->nh(), nmean = mean_param_count(h->mean_kind, d), m = (int)G.size();
    Workspace ws = h->bws;
    Ctx cx = h->cx();
    cx.no_workers = true;
    HIPCHK(hipMemcpyAsync(h->bzmap, pos.data(), m * sizeof(int), hipMemcpyHostToDevice, h->stream));
    h->train_flop += m * ((double)h->N * h->N * h->N / 3.0) * (h->lock_inv_panels > 0 ? 2.0 : 1.0);   // (L^-1 now) + K^-1
    if (h->lock_inv_panels > 0) {                 // the value batch stopped at L and the diagonal blocks' inverses
        PhaseTimer t(h, GPMPC_PH_FACTOR);
        twolevel_inverse_all(cx, ws, cx.stream, h->lock_inv_panels, m, h->bzmap);
    }
    {
        PhaseTimer t(h, GPMPC_PH_SOLVE);
        solve_alpha_from_w(cx, ws, m, h->bzmap);
    }
    {
        PhaseTimer t(h, GPMPC_PH_INVK);
        CHK(invk_lower(cx, ws, m, h->bzmap, pos.data()));
    }
    {
        PhaseTimer t(h, GPMPC_PH_NLL);
        hipLaunchKernelGGL(nll_grad_kernel, dim3(Np / 64, Np / 64, m), dim3(256), 0, cx.stream, h->XT, ws.hyper, ws.InvK, ws.alpha,
                           h->bgradPartial, h->N, Np, d, (const int*)h->bzmap);
        hipLaunchKernelGGL(nll_grad_finish_kernel, dim3(m), dim3(256), 0, cx.stream, h->bgradPartial, ws.hyper, h->bgradOut, Np, d,
                           BGS, (const int*)h->bzmap);
        if (nmean)
            hipLaunchKernelGGL(mean_grad_kernel, dim3(m), dim3(256), 0, cx.stream, h->XT, ws.alpha, h->bgradOut + d + 2, h->mean_kind,
                               h->N, Np, d, BGS, (const int*)h->bzmap);
    }
    HIPCHK(hipGetLastError());
    std::vector<double> gv((size_t)h->bws.batch * BGS);
    HIPCHK(hipMemcpyAsync(gv.data(), h->bgradOut, gv.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < m; ++i) {
        std::memcpy(G[i]->g, &gv[(size_t)pos[i] * BGS], nh * sizeof(double));
        G[i]->rc = GPMPC_OK;
        add_log_prior(h, G[i]->theta, nullptr, G[i]->g);
    }
    return GPMPC_OK;
}

// Evaluates every request (any number, mixed).  Device failures are returned; unusable points are marked in their request.
// retain: the value-only points of this call fit one batch and their factors are to be remembered per restart (h->lock_ret)
// for a grad_of_last request in the NEXT call.
static int nll_batch(gpmpc_gp* h, int a, std::vector<NllReq*>& reqs, bool retain = false) {
    const int d = h->d, nh = h->nh();
    std::vector<NllReq*> group[2], last;
    std::vector<int> last_pos;
    for (NllReq* r : reqs) {
        if (r->grad_of_last) {
            r->rc = GPMPC_EINVAL;                // unless the slot still holds this point
            if (r->id >= 0 && r->id < (int)h->lock_ret.size()) {
                const auto& lr = h->lock_ret[r->id];
                if (lr.pos >= 0 && (int)lr.theta.size() == nh && std::memcmp(lr.theta.data(), r->theta, nh * sizeof(double)) == 0) {
                    last.push_back(r);
                    last_pos.push_back(lr.pos);
                }
            }
            continue;
        }
        bool usable = true;
        for (int k = 0; k < d + 1; ++k) usable &= (r->theta[k] == r->theta[k]) && r->theta[k] != 0.0;
        for (int k = d + 1; k < nh; ++k) usable &= r->theta[k] == r->theta[k];
        if (!usable) { r->rc = GPMPC_EINVAL; continue; }
        if (g_fail_nll_after.load(std::memory_order_relaxed) > 0 && g_fail_nll_after.fetch_sub(1) == 1)
            return fail(GPMPC_EHIP, "injected device failure (fail_nll_after)");
        r->rc = GPMPC_ENOTPD;                    // until an evaluation succeeds
        group[r->want_grad ? 1 : 0].push_back(r);
    }
    static const bool verbose = getenv("GPMPC_VERBOSE") != nullptr;
    if (!last.empty()) {                         // first: the batches below overwrite the slots
        const auto t0 = std::chrono::steady_clock::now();
        CHK(nll_grad_retained(h, a, last, last_pos));
        if (verbose)
            fprintf(stderr, "gpmpc: lock-step gradients of %d retained point%s (%s): %.3f ms\n", (int)last.size(), last.size() == 1 ? "" : "s",
                    h->lock_inv_panels > 0 ? "L^-1 formed now" : "L^-1 was there",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    const size_t total = group[0].size() + group[1].size();
    if (total == 0) return GPMPC_OK;
    for (auto& lr : h->lock_ret) lr.pos = -1;
    CHK(ensure_batch_ws(h, (int)std::max(group[0].size(), group[1].size())));
    const int cap = h->bws.batch;
    for (int wg = 1; wg >= 0; --wg) {            // value-only points last: their factors stay in the slots
        std::vector<NllReq*>& G = group[wg];
        for (size_t b0 = 0; b0 < G.size(); b0 += cap) {
            const int n = (int)std::min<size_t>(cap, G.size() - b0);
            std::vector<int> failed;
            const auto t0 = std::chrono::steady_clock::now();
            static const bool skip_inv = !(getenv("GPMPC_TRAIN_SKIP_INVERSE") && atoi(getenv("GPMPC_TRAIN_SKIP_INVERSE")) == 0);
            const bool vonly = wg == 0 && retain && skip_inv && G.size() <= (size_t)cap;
            CHK(nll_batch_core(h, a, n, &G[b0], wg == 1, 0.0, failed, vonly));
            const int inv_first = h->lock_inv_panels;   // what this pass left in Inv (one value for the whole batch workspace)
            bool inv_mixed = false;
            if (verbose)
                fprintf(stderr, "gpmpc: lock-step batch of %d point%s (%s): %.3f ms, %d to repeat with jitter\n", n, n == 1 ? "" : "s",
                        wg ? "value + gradient" : "value", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                        (int)failed.size());
            std::vector<int> slot_of(n);          // where each point's factors are now
            for (int i = 0; i < n; ++i) slot_of[i] = i;
            if (!failed.empty()) {                // the reference's one-shot jitter, for the points that need it
                std::vector<NllReq*> again;
                for (int i : failed) again.push_back(G[b0 + i]);
                std::vector<int> failed2;
                CHK(nll_batch_core(h, a, (int)again.size(), again.data(), wg == 1, 1e-8, failed2, vonly));
                // the repeat may have taken another execution than the first pass (a hand-off time-out, a parked chain): the
                // slots then hold two kinds of Inv and h->lock_inv_panels describes only the repeat's -- retain nothing,
                // the restarts re-evaluate (train_native.hpp: a refused grad_of_last is a full evaluation)
                inv_mixed = h->lock_inv_panels != inv_first;
                // (the repeat ran in the first slots: what was there is gone, the repeated points now live there)
                const int m = (int)again.size();
                for (int i = 0; i < m && i < n; ++i) slot_of[i] = -1;
                for (int j = 0; j < m; ++j) slot_of[failed[j]] = j;
            }
            if (wg == 0 && retain && !inv_mixed && G.size() <= (size_t)cap) {
                for (int i = 0; i < n; ++i) {
                    NllReq* r = G[i];
                    if (r->rc != GPMPC_OK || slot_of[i] < 0 || r->id < 0 || r->id >= (int)h->lock_ret.size()) continue;
                    h->lock_ret[r->id].pos = slot_of[i];
                    h->lock_ret[r->id].theta.assign(r->theta, r->theta + nh);
                }
            }
        }
    }
    // completion on the host, as gpmpc_nll does: the hyper-priors (calc_NLL has no N/2 log 2 pi term)
    for (NllReq* r : reqs)
        if (r->rc == GPMPC_OK && !r->grad_of_last) add_log_prior(h, r->theta, &r->f, r->want_grad ? r->g : nullptr);
    h->nll_last_a = -1;
    return GPMPC_OK;
}

// One restart's search as a thread that parks at every evaluation; the driver runs the parked requests as batches.
struct LockstepPool {
    std::mutex m;
    std::condition_variable cv_driver, cv_worker;
    int n_threads = 0, n_parked = 0, n_done = 0;
    long epoch = 0;
    bool abort = false;                          // a device failure: every further evaluation is refused
    struct Slot {
        NllReq req;
        std::vector<double> theta, grad;
        bool posted = false;
        long served_epoch = -1;
    };
    std::vector<Slot> slots;
    // worker side: evaluate `theta` (value, or value + gradient into g); false = unusable point / failure
    // (of_last: no evaluation, the gradient of the point this restart evaluated last -- value only -- into g)
    bool evaluate(int id, const double* theta, int nh, double* f, double* g, bool of_last = false) {
        std::unique_lock<std::mutex> lk(m);
        if (abort) return false;
        Slot& s = slots[id];
        s.theta.assign(theta, theta + nh);
        s.grad.assign(nh, 0.0);
        s.req = NllReq();
        s.req.theta = s.theta.data();
        s.req.id = id;
        s.req.grad_of_last = of_last;
        s.req.want_grad = g != nullptr;
        s.req.g = s.grad.data();
        s.posted = true;
        ++n_parked;
        const long my_epoch = epoch;
        cv_driver.notify_one();
        cv_worker.wait(lk, [&] { return s.served_epoch >= my_epoch && !s.posted; });
        if (s.req.rc != GPMPC_OK) return false;
        if (f) *f = s.req.f;
        if (g) std::memcpy(g, s.grad.data(), nh * sizeof(double));
        return true;
    }
    void finished() {
        std::lock_guard<std::mutex> lk(m);
        ++n_done;
        cv_driver.notify_one();
    }
};

// Which HIP runtime and which RCCL this process runs the library on (the library is linked against libamdhip64.so.N by
// soname and takes whichever the process mapped first -- PyTorch's bundled copy when torch was imported first, /opt/rocm's
// otherwise; RCCL is bound at run time to the librccl next to that HIP runtime, train_native.hpp).  Text: "hip_runtime=<version> hip_path=<file>
// rccl=<version|unavailable> rccl_path=<file>".
extern "C" int gpmpc_runtime_info(char* buf, int buflen) {
    if (!buf || buflen <= 0) return fail(GPMPC_EINVAL, "bad buffer");
    int hv = 0;
    std::string hip_path = "?";
#ifndef GPMPC_EMULATED
    (void)hipRuntimeGetVersion(&hv);
    Dl_info di;
    if (dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) hip_path = di.dli_fname;
#else
    hip_path = "emulator";
#endif
    RcclApi& R = rccl_api();
    int rv = 0;
    if (R.ok() && R.GetVersion) (void)R.GetVersion(&rv);
    char rver[32];
    if (R.ok()) snprintf(rver, sizeof(rver), "%d", rv); else snprintf(rver, sizeof(rver), "unavailable");
    snprintf(buf, buflen, "hip_runtime=%d hip_path=%s rccl=%s rccl_path=%s", hv, hip_path.c_str(), rver, R.ok() ? R.path.c_str() : "-");
    return GPMPC_OK;
}

extern "C" int gpmpc_train_multistart(gpmpc_gp* h, int nstart, const double* starts, const double* lb, const double* ub,
                                      int max_iter, double tol, int rank, int world, void* rccl_comm, int want_invK,
                                      double* hyper_opt, double* obj, double* theta_all, int* info, int* status) {
    if (!h || nstart <= 0 || !starts || !lb || !ub || !hyper_opt) return fail(GPMPC_EINVAL, "NULL argument or nstart <= 0");
    if (world < 1 || rank < 0 || rank >= world) return fail(GPMPC_EINVAL, "bad rank %d / world %d", rank, world);
    HIPCHK(hipSetDevice(h->device));
    const int Ny = h->Ny, nh = h->nh(), d = h->d, row = nh + 1;
    const double inf = std::numeric_limits<double>::infinity();
    if (max_iter <= 0) max_iter = 200;
    if (!(tol > 0.0)) tol = 1e-8;
    // [a][r][NLL, theta...]; not-owned / failed: +inf.  One more word travels with the table: this rank's status (0 = fine,
    // else the GPMPC_E* code of a device failure).  A rank that fails does NOT return before the exchange -- its peers
    // would wait in the collective for ever -- it sends +inf rows and its code; every rank then returns that error.
    std::vector<double> table((size_t)Ny * nstart * row + 1, 0.0);
    int local_rc = GPMPC_OK;
    std::string local_err;
    long iters_total = 0, evals_total = 0;
    h->train_flop = 0.0;
    for (int a = 0; a < Ny && local_rc == GPMPC_OK; ++a) {
        BoxProblem P;
        P.n = nh;
        P.lb.assign(lb + (size_t)a * nh, lb + (size_t)(a + 1) * nh);
        P.ub.assign(ub + (size_t)a * nh, ub + (size_t)(a + 1) * nh);
        P.logv.resize(nh);
        for (int k = 0; k < nh; ++k) {
            // (argument errors are the same on every rank: returning here cannot strand a peer)
            if (!(P.lb[k] <= P.ub[k])) return fail(GPMPC_EINVAL, "empty box for hyper-parameter %d of output %d", k, a);
            // length scales and sf in log space (their boxes span many decades).  NOT the noise sn: the NLL sees it as sn^2,
            // so d NLL / d log sn = 2 sn^2 (...) vanishes at the reference's start sn = 1e-5 and a log-space search leaves it
            // there -- on the fixture whose optimum has sn on its upper bound it stopped 5.6 above the reference's NLL.
            P.logv[k] = k < d + 1 && P.lb[k] > 0.0 && P.ub[k] < inf;
        }
        P.eval = [&](const double* th, double* f, double* g) -> bool {
            if (local_rc != GPMPC_OK) return false;              // after a device failure: every point is unusable
            const int rc = gpmpc_nll(h, a, th, f, g, nullptr);
            if (rc == GPMPC_EHIP || rc == GPMPC_ENOMEM) { local_rc = rc; local_err = g_err; }
            return rc == GPMPC_OK;
        };
        P.grad_last = [&](const double* th, double* g) -> bool {
            if (local_rc != GPMPC_OK) return false;
            const int rc = nll_grad_last(h, a, th, g);
            if (rc == GPMPC_EHIP || rc == GPMPC_ENOMEM) { local_rc = rc; local_err = g_err; }
            return rc == GPMPC_OK;
        };
        // one restart: the two-stage search (train_native.hpp) from its start; counts go to the totals
        auto run_restart = [&](const BoxProblem& Pr, int r, long& iters, long& evals) -> BoxResult {
            BoxResult res = minimize_box_lbfgs(Pr, starts + ((size_t)a * nstart + r) * nh, max_iter, tol);
            iters += res.iters; evals += res.evals;
            // The linear noise variable is badly scaled against the log variables (its whole box is 1e-2 wide): once the
            // first search has stopped with iterations to spare, a second one from there with sn in log space -- where its
            // gradient no longer vanishes -- polishes the optimum (third reference-made fixture: -95.7 -> the -197.7 that
            // SLSQP with the analytic gradient finds; the reference's own run stops at -80.3).
            if (res.ok && res.iters < max_iter && Pr.lb[d + 1] > 0.0 && Pr.ub[d + 1] < inf && local_rc == GPMPC_OK) {
                BoxProblem P2 = Pr;
                P2.logv[d + 1] = 1;
                const BoxResult res2 = minimize_box_lbfgs(P2, res.theta.data(), max_iter - res.iters, tol);
                iters += res2.iters; evals += res2.evals;
                if (res2.ok && res2.f < res.f) res = res2;
            }
            return res;
        };
        std::vector<int> mine;
        for (int r = 0; r < nstart; ++r) {
            table[((size_t)a * nstart + r) * row] = inf;
            if (r % world == rank) mine.push_back(r);
        }
        // Lock-step: the restarts of this rank advance together and their evaluation points form batches (above).  `nstart`
        // is the same on every rank, so the choice is too.  GPMPC_TRAIN_LOCKSTEP=0 / a single restart: one after the other.
        static const bool lockstep_env = !(getenv("GPMPC_TRAIN_LOCKSTEP") && atoi(getenv("GPMPC_TRAIN_LOCKSTEP")) == 0);
        if (lockstep_env && nstart > 1 && !mine.empty() && local_rc == GPMPC_OK) {
            LockstepPool pool;
            pool.n_threads = (int)mine.size();
            pool.slots.resize(mine.size());
            // value-only trials with the gradient of the accepted point from its retained factors: when this rank's restarts
            // fit one batch (GPMPC_TRAIN_RETAIN=0: every trial with its gradient, as when they do not fit)
            static const bool retain_env = !(getenv("GPMPC_TRAIN_RETAIN") && atoi(getenv("GPMPC_TRAIN_RETAIN")) == 0);
            bool retain = false;
            if (retain_env) {
                const int erc = ensure_batch_ws(h, (int)mine.size());
                if (erc != GPMPC_OK) { local_rc = erc; local_err = g_err; }
                retain = erc == GPMPC_OK && h->bws.batch >= (int)mine.size();
            }
            h->lock_ret.assign(mine.size(), gpmpc_gp::LockRet());
            std::vector<BoxResult> results(mine.size());
            std::vector<long> it_each(mine.size(), 0), ev_each(mine.size(), 0);
            std::vector<std::thread> threads;
            for (int id = 0; id < (int)mine.size(); ++id)
                threads.emplace_back([&, id]() {
                    BoxProblem Pt = P;
                    if (retain) {
                        Pt.eval = [&pool, id, nh](const double* th, double* f, double* g) -> bool { return pool.evaluate(id, th, nh, f, g); };
                        Pt.grad_last = [&pool, id, nh](const double* th, double* g) -> bool { return pool.evaluate(id, th, nh, nullptr, g, true); };
                    } else {                                    // (value + gradient at every trial point: one kind of batch)
                        Pt.eval = [&pool, id, nh](const double* th, double* f, double* g) -> bool {
                            std::vector<double> gtmp(nh);
                            return pool.evaluate(id, th, nh, f, g ? g : gtmp.data());
                        };
                        Pt.grad_last = nullptr;
                    }
                    results[id] = run_restart(Pt, mine[id], it_each[id], ev_each[id]);
                    pool.finished();
                });
            {
                std::unique_lock<std::mutex> lk(pool.m);
                for (;;) {
                    pool.cv_driver.wait(lk, [&] { return pool.n_parked + pool.n_done == pool.n_threads; });
                    if (pool.n_done == pool.n_threads) break;
                    std::vector<NllReq*> reqs;
                    for (auto& sl : pool.slots)
                        if (sl.posted) reqs.push_back(&sl.req);
                    lk.unlock();
                    int rc = local_rc == GPMPC_OK ? nll_batch(h, a, reqs, retain) : local_rc;
                    lk.lock();
                    if (rc != GPMPC_OK && local_rc == GPMPC_OK) { local_rc = rc; local_err = g_err; }
                    if (local_rc != GPMPC_OK) {
                        pool.abort = true;                      // (every parked and every later evaluation is refused)
                        for (NllReq* rq : reqs) rq->rc = local_rc;
                    }
                    for (auto& sl : pool.slots)
                        if (sl.posted) { sl.posted = false; sl.served_epoch = pool.epoch; }
                    pool.n_parked = 0;
                    ++pool.epoch;
                    pool.cv_worker.notify_all();
                }
            }
            for (auto& t : threads) t.join();
            for (int id = 0; id < (int)mine.size(); ++id) {
                double* out = &table[((size_t)a * nstart + mine[id]) * row];
                iters_total += it_each[id]; evals_total += ev_each[id];
                std::memcpy(out + 1, results[id].theta.data(), nh * sizeof(double));
                if (results[id].ok && local_rc == GPMPC_OK) out[0] = results[id].f;
            }
        } else {
            for (int r : mine) {
                double* out = &table[((size_t)a * nstart + r) * row];
                if (local_rc != GPMPC_OK) continue;
                BoxResult res = run_restart(P, r, iters_total, evals_total);
                std::memcpy(out + 1, res.theta.data(), nh * sizeof(double));
                if (res.ok && local_rc == GPMPC_OK) out[0] = res.f;
            }
        }
    }
    h->train_iters = iters_total;
    h->train_evals = evals_total;
    if (local_rc != GPMPC_OK)                                   // whatever this rank found is not to be trusted
        for (size_t e = 0; e + 1 < table.size(); e += row) table[e] = inf;
    table.back() = (double)local_rc;
    const size_t cnt = table.size();
    if (rccl_comm) {    // one all-gather of the whole table: (1 + nh) doubles per restart (also at world = 1: a self-gather)
        RcclApi& R = rccl_api();
        // (a missing librccl is the same on every rank of a node, and the communicator could not exist without it)
        if (!R.ok()) return fail(GPMPC_EHIP, "librccl.so could not be loaded: %s", R.load_error.c_str());
        struct DevBuf {                                            // freed on every path
            double* p = nullptr;
            ~DevBuf() { if (p) (void)hipFree(p); }
        } dsend, drecv;
        std::vector<double> all(cnt * world);
        hipError_t he = hipMalloc(&dsend.p, cnt * sizeof(double));
        if (he == hipSuccess) he = hipMalloc(&drecv.p, cnt * world * sizeof(double));
        if (he == hipSuccess) he = hipMemcpyAsync(dsend.p, table.data(), cnt * sizeof(double), hipMemcpyHostToDevice, h->stream);
        // A rank that cannot even stage its table cannot join the collective; nothing this library can do would free its
        // peers then (they time out in RCCL).  With a few KB per rank that means the device is gone.
        if (he != hipSuccess) return fail(GPMPC_EHIP, "staging the restart table failed: %s", hipGetErrorString(he));
        (void)hipGetLastError();
        const int rc = R.AllGather(dsend.p, drecv.p, cnt, RCCL_FLOAT64, rccl_comm, h->stream);
        if (rc != 0) return fail(GPMPC_EHIP, "ncclAllGather failed: %s", R.GetErrorString ? R.GetErrorString(rc) : "?");
        HIPCHK(hipMemcpyAsync(all.data(), drecv.p, all.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int q = 0; q < world; ++q) {
            const int peer_rc = (int)all[(size_t)q * cnt + cnt - 1];
            if (peer_rc != GPMPC_OK) {
                if (q == rank) return fail(peer_rc, "%s", local_err.c_str());
                return fail(peer_rc, "rank %d of the restart shard reported a device failure (code %d)", q, peer_rc);
            }
        }
        for (int a = 0; a < Ny; ++a)
            for (int r = 0; r < nstart; ++r)
                std::memcpy(&table[((size_t)a * nstart + r) * row], &all[(size_t)(r % world) * cnt + ((size_t)a * nstart + r) * row],
                            row * sizeof(double));
    } else if (local_rc != GPMPC_OK && world == 1) {
        return fail(local_rc, "%s", local_err.c_str());
    }
    // (world > 1 without a communicator: the caller merges the ranks' tables; a failed rank reports through `status`)
    if (status) *status = local_rc;
    const bool merged = world == 1 || rccl_comm != nullptr;
    bool all_ok = true;
    for (int a = 0; a < Ny; ++a) {
        int best = -1;
        for (int r = 0; r < nstart; ++r) {
            const double* e = &table[((size_t)a * nstart + r) * row];
            if (obj) obj[(size_t)a * nstart + r] = e[0];
            if (theta_all) std::memcpy(theta_all + ((size_t)a * nstart + r) * nh, e + 1, nh * sizeof(double));
            if (e[0] < inf && (best < 0 || e[0] < table[((size_t)a * nstart + best) * row])) best = r;   // first minimum: np.argmin
        }
        if (best >= 0) std::memcpy(hyper_opt + (size_t)a * nh, &table[((size_t)a * nstart + best) * row + 1], nh * sizeof(double));
        else all_ok = false;
    }
    if (!merged) {                                              // caller merges the ranks' tables and calls gpmpc_fit
        if (local_rc != GPMPC_OK) g_err = local_err;            // (text for gpmpc_last_error; the code is in *status)
        return GPMPC_OK;
    }
    if (!all_ok) return fail(GPMPC_ENOTPD, "every restart of an output failed (K not positive definite along the way)");
    return gpmpc_fit(h, hyper_opt, want_invK, info);            // optimize.py:476-494 at theta*
}

