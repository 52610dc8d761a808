// api_rollout.inl -- part of gpmpc_api.hip (one translation unit; included in order, not compiled alone).
// Concern: T-step propagation on the device (open loop / state feedback), hipGraph replay.
static int rollout_impl(gpmpc_gp* h, int method, int T, const double* z0, const double* U, const double* Sigma0,
                        const double* sa, const double* sb, const double* Kz, const double* k0, const double* Kc,
                        double* mean, double* cov, double* Uout) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    if (!h->fitted) return fail(GPMPC_ENOTFIT, "model has no factors (call gpmpc_fit or gpmpc_set_factors)");
    if (method < GPMPC_ME || method > GPMPC_OLD_TA) return fail(GPMPC_EINVAL, "No GP method with code %d", method);
    const int d = h->d, Ny = h->Ny, Nu = d - Ny;
    const bool fb = Kz != nullptr;
    if (T <= 0 || !z0 || !Sigma0 || !mean || !cov || (!fb && Nu > 0 && !U)) return fail(GPMPC_EINVAL, "bad T or NULL argument");
    if (Nu < 0) return fail(GPMPC_EINVAL, "roll-out needs d >= Ny (inputs are [state, control])");
    if (fb && (Nu == 0 || !k0 || !Kc)) return fail(GPMPC_EINVAL, "feedback roll-out needs controls and Kz, k0, Kc");
    if (method == GPMPC_OLD_TA && h->mean_kind)
        return fail(GPMPC_EINVAL, "'old_TA' with a non-zero mean function raises in the reference (gp_functions.py:309-311); not served");
    HIPCHK(hipSetDevice(h->device));
    CHK(ensure_scratch(h, 1));
    const bool moments = method == GPMPC_EM || method == GPMPC_OLD_ME || method == GPMPC_OLD_TA;
    if (moments && !h->have_invK) {
        CHK(compute_invK(h->cx(), h->ws));
        h->have_invK = true;
    }
    // device staging (grow-only, with a pinned mirror): inputs [z | Sigma | sa | sb | Kz | k0 | Kc | U], then the
    // trajectories [mean (T) | cov (T)] and scratch [var | J]: one copy up, one copy down ([U |] mean | cov)
    const int nu1 = std::max(Nu, 1);
    const size_t nz = d, nS = (size_t)d * d, nU = (size_t)T * nu1, nM = (size_t)T * Ny, nC = (size_t)T * Ny * Ny;
    const size_t nK = (size_t)nu1 * Ny;
    const size_t nIn = nz + nS + 2 * Ny + 2 * nK + nu1 + nU;
    const size_t total = nIn + nM + nC + Ny + (size_t)Ny * d;
    if (total > h->roll_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        drop_roll_graphs(h);
        hipFree(h->roll_dev);
        if (h->roll_pin) hipHostFree(h->roll_pin);
        h->roll_dev = h->roll_pin = nullptr;
        h->roll_cap = 0;
        HIPCHK(hipMalloc(&h->roll_dev, total * sizeof(double)));
        HIPCHK(hipHostMalloc((void**)&h->roll_pin, total * sizeof(double), hipHostMallocDefault));
        h->roll_cap = total;
    }
    double* buf = h->roll_dev;
    double *dz = buf, *dS = dz + nz, *dsa = dS + nS, *dsb = dsa + Ny, *dKz = dsb + Ny, *dk0 = dKz + nK, *dKc = dk0 + nu1,
           *dU = dKc + nK, *dM = dU + nU, *dC = dM + nM, *dV = dC + nC, *dJ = dV + Ny;
    {
        double* pz = h->roll_pin;
        auto put = [&](double* dev_dst, const double* src, size_t n) { std::memcpy(pz + (dev_dst - buf), src, n * sizeof(double)); };
        std::memset(pz, 0, nIn * sizeof(double));
        put(dz, z0, nz);
        put(dS, Sigma0, nS);
        for (int a = 0; a < Ny; ++a) pz[(dsa - buf) + a] = sa ? sa[a] : 1.0;
        if (sb) put(dsb, sb, Ny);
        if (!fb && Nu > 0) put(dU, U, (size_t)T * Nu);
        if (fb) {
            put(dKz, Kz, nK);
            put(dk0, k0, Nu);
            put(dKc, Kc, nK);
            put(dU, z0 + Ny, Nu);                                  // the first control comes with z0
        }
    }
    HIPCHK(hipMemcpyAsync(buf, h->roll_pin, nIn * sizeof(double), hipMemcpyHostToDevice, h->stream));
    int rc = GPMPC_OK;
    auto enqueue_steps = [&]() -> int {
        int r = GPMPC_OK;
        for (int t = 0; t < T && r == GPMPC_OK; ++t) {
            if (t > 0)
                hipLaunchKernelGGL(rollout_feed_kernel, dim3(1), dim3(64), 0, h->stream, dM + (size_t)(t - 1) * Ny,
                                   dC + (size_t)(t - 1) * Ny * Ny, dU + (size_t)t * nu1, dsa, dsb, dz, dS, Ny, d,
                                   fb ? dKz : (const double*)nullptr, fb ? dk0 : (const double*)nullptr,
                                   fb ? dKc : (const double*)nullptr, fb ? dU + (size_t)t * nu1 : (double*)nullptr);
            double* oM = dM + (size_t)t * Ny;
            double* oC = dC + (size_t)t * Ny * Ny;
            if (moments) {
                r = predict_moments_chunk(h, method, 1, dz, dS, oM, oC);
            } else {
                const bool ta = method == GPMPC_TA;
                r = predict_chunk(h, 1, dz, oM, dV, ta ? dJ : nullptr);
                if (r == GPMPC_OK)
                    hipLaunchKernelGGL(cov_assemble_kernel, dim3((unsigned)((Ny * Ny + 255) / 256)), dim3(256), 0, h->stream, dV, dJ,
                                       ta ? dS : (const double*)nullptr, oC, 1, Ny, d);
            }
        }
        return r;
    };
    // The loop is 6-9 dependent launches per step: at the reference's model sizes that is all the time there is (26 us per
    // 'ME' step at N = 200).  After one plain run with the same key -- every lazy allocation and kernel attribute is then in
    // place -- the T-step loop is captured into a hipGraph and replayed; the key holds every address and size a launch bakes in.
    bool done = false;
    if (moments) CHK(ensure_beta(h));        // lazily refreshed after a fit: must not hide inside (or be missing from) a captured loop
#ifndef GPMPC_EMULATED
    static const bool use_graph = !(getenv("GPMPC_ROLLOUT_GRAPH") && atoi(getenv("GPMPC_ROLLOUT_GRAPH")) == 0);
    static const int graph_max_np = getenv("GPMPC_ROLLOUT_GRAPH_NP") ? atoi(getenv("GPMPC_ROLLOUT_GRAPH_NP")) : 2048;
    if (use_graph && !h->prof.on && h->Np <= graph_max_np) {
        const std::vector<long> key = {method, T, fb ? 1 : 0, Nu, h->N, h->Np, Ny, d, h->mean_kind, h->mean_add ? 1 : 0, h->Bcap,
                                       (long)buf, (long)h->XT, (long)h->ws.hyper, (long)h->ws.alpha, (long)h->ws.Inv,
                                       (long)h->ws.InvK, (long)h->beta, (long)h->KsT, (long)h->meanT, (long)h->part,
                                       (long)h->ccpart, (long)h->em, h->emBytes, (long)h->UT, (long)h->mpar, (long)h->stream};
        gpmpc_gp::RollGraph* g = nullptr;
        for (auto& e : h->roll_graphs)
            if (e.key == key) g = &e;
        if (!g && h->roll_warm == key) {
            gpmpc_gp::RollGraph ng;
            if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int r = enqueue_steps();
                const hipError_t ee = hipStreamEndCapture(h->stream, &ng.graph);
                if (r == GPMPC_OK && ee == hipSuccess && ng.graph && hipGraphInstantiate(&ng.exec, ng.graph, nullptr, nullptr, 0) == hipSuccess) {
                    ng.key = key;
                    if (h->roll_graphs.size() >= 8) drop_roll_graphs(h);
                    h->roll_graphs.push_back(ng);
                    g = &h->roll_graphs.back();
                } else {
                    if (ng.graph) hipGraphDestroy(ng.graph);
                    (void)hipGetLastError();
                }
            }
        }
        if (g) {
            HIPCHK(hipGraphLaunch(g->exec, h->stream));
            done = true;
        } else {
            h->roll_warm = key;
        }
    }
#endif
    if (!done) rc = enqueue_steps();
    if (rc == GPMPC_OK) {
        const bool wantU = Uout && Nu > 0;
        double* first = wantU ? dU : dM;
        hipError_t e = hipMemcpyAsync(h->roll_pin + (first - buf), first, ((wantU ? nU : 0) + nM + nC) * sizeof(double),
                                      hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) return fail(GPMPC_EHIP, "%s", hipGetErrorString(e));
        std::memcpy(mean, h->roll_pin + (dM - buf), nM * sizeof(double));
        std::memcpy(cov, h->roll_pin + (dC - buf), nC * sizeof(double));
        if (wantU) {
            if (nu1 == Nu) std::memcpy(Uout, h->roll_pin + (dU - buf), (size_t)T * Nu * sizeof(double));
        }
    } else {
        hipStreamSynchronize(h->stream);
    }
    return rc;
}

extern "C" int gpmpc_rollout(gpmpc_gp* h, int method, int T, const double* z0, const double* U, const double* Sigma0,
                             const double* sa, const double* sb, double* mean, double* cov) {
    return rollout_impl(h, method, T, z0, U, Sigma0, sa, sb, nullptr, nullptr, nullptr, mean, cov, nullptr);
}

extern "C" int gpmpc_rollout_feedback(gpmpc_gp* h, int method, int T, const double* z0, const double* Sigma0, const double* sa,
                                      const double* sb, const double* Kz, const double* k0, const double* Kc, double* mean,
                                      double* cov, double* U_out) {
    if (!Kz) return fail(GPMPC_EINVAL, "Kz is NULL");
    return rollout_impl(h, method, T, z0, nullptr, Sigma0, sa, sb, Kz, k0, Kc, mean, cov, U_out);
}


// a17, "batch across trajectories / methods" (gp_class.py:777-804 loops over the methods and, inside, over the
// steps): M trajectories -- each with its own method, start, controls and initial input covariance -- advance in LOCK-STEP.
// Per time step ONE pass over the factors serves all of them: the 'ME' / 'TA' trajectories form one prediction batch (one
// stream of the lower triangles of L^-1, 4 N (N+1) bytes per output, whatever their number up to 32: the HBM-bound part of
// a roll-out at C3 size), the trajectories of a moment method one batched launch set.  Trajectories are independent: a
// trajectory's numbers do not depend on what else is in the call, with one exception spelt out in include/gpmpc.h (a lone
// 'ME' / 'TA' trajectory takes the one-column variance kernel, two or more the batched one: two summation orders).
extern "C" int gpmpc_rollout_multi(gpmpc_gp* h, int M, const int* methods, int T, const double* z0, const double* U,
                                   const double* Sigma0, const double* sa, const double* sb, double* mean, double* cov) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    if (!h->fitted) return fail(GPMPC_ENOTFIT, "model has no factors (call gpmpc_fit or gpmpc_set_factors)");
    const int d = h->d, Ny = h->Ny, Nu = d - Ny;
    if (M <= 0 || M > 64 || T <= 0 || !methods || !z0 || !Sigma0 || !mean || !cov || (Nu > 0 && !U))
        return fail(GPMPC_EINVAL, "bad M (1..64), T or NULL argument");
    if (Nu < 0) return fail(GPMPC_EINVAL, "roll-out needs d >= Ny (inputs are [state, control])");
    for (int m = 0; m < M; ++m) {
        if (methods[m] < GPMPC_ME || methods[m] > GPMPC_OLD_TA) return fail(GPMPC_EINVAL, "No GP method with code %d", methods[m]);
        if (methods[m] == GPMPC_OLD_TA && h->mean_kind)
            return fail(GPMPC_EINVAL, "'old_TA' with a non-zero mean function raises in the reference (gp_functions.py:309-311); not served");
    }
    HIPCHK(hipSetDevice(h->device));
    CHK(ensure_scratch(h, M));
    // device order: 'ME', 'TA', then the moment methods, each group contiguous (perm[k] = caller's index of device slot k)
    std::vector<int> perm;
    int cnt[5] = {0, 0, 0, 0, 0};
    for (int code : {GPMPC_ME, GPMPC_TA, GPMPC_EM, GPMPC_OLD_ME, GPMPC_OLD_TA})
        for (int m = 0; m < M; ++m)
            if (methods[m] == code) { perm.push_back(m); ++cnt[code]; }
    const int nME = cnt[GPMPC_ME], nTA = cnt[GPMPC_TA], nA = nME + nTA;
    const bool moments = nA < M;
    const int nu1 = std::max(Nu, 1);
    const size_t nZ = (size_t)M * d, nS = (size_t)M * d * d, nU = (size_t)T * M * nu1, nM = (size_t)T * M * Ny, nC = (size_t)T * M * Ny * Ny;
    const size_t nIn = nZ + nS + 2 * Ny + nU;
    const size_t total = nIn + nM + nC + (size_t)M * Ny + (size_t)M * Ny * d;
    if (total > h->rollm_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        hipFree(h->rollm_dev);
        if (h->rollm_pin) hipHostFree(h->rollm_pin);
        h->rollm_dev = h->rollm_pin = nullptr;
        h->rollm_cap = 0;
        HIPCHK(hipMalloc(&h->rollm_dev, total * sizeof(double)));
        HIPCHK(hipHostMalloc((void**)&h->rollm_pin, total * sizeof(double), hipHostMallocDefault));
        h->rollm_cap = total;
    }
    double* buf = h->rollm_dev;
    double *dZ = buf, *dS = dZ + nZ, *dsa = dS + nS, *dsb = dsa + Ny, *dU = dsb + Ny, *dM = dU + nU, *dC = dM + nM, *dV = dC + nC,
           *dJ = dV + (size_t)M * Ny;
    {
        double* pz = h->rollm_pin;
        std::memset(pz, 0, nIn * sizeof(double));
        for (int k = 0; k < M; ++k) {
            const int m = perm[k];
            std::memcpy(pz + (dZ - buf) + (size_t)k * d, z0 + (size_t)m * d, d * sizeof(double));
            std::memcpy(pz + (dS - buf) + (size_t)k * d * d, Sigma0 + (size_t)m * d * d, (size_t)d * d * sizeof(double));
            for (int t = 0; t < T && Nu > 0; ++t)     // time-major on the device: the controls of step t are one block
                std::memcpy(pz + (dU - buf) + ((size_t)t * M + k) * nu1, U + ((size_t)m * T + t) * Nu, Nu * sizeof(double));
        }
        for (int a = 0; a < Ny; ++a) pz[(dsa - buf) + a] = sa ? sa[a] : 1.0;
        if (sb) std::memcpy(pz + (dsb - buf), sb, Ny * sizeof(double));
    }
    HIPCHK(hipMemcpyAsync(buf, h->rollm_pin, nIn * sizeof(double), hipMemcpyHostToDevice, h->stream));
    // Trajectories are independent, so the two groups need not even meet per time step: with both kinds in the call the moment
    // methods' whole horizon goes to the workers' queue -- K^-1 first if the model does not have it yet (a fit without it: the
    // 'ME' / 'TA' group then runs NEXT TO that product, MFMA-bound against HBM-bound), then their T steps -- while the 'ME' / 'TA'
    // group's T steps run on the main queue; the queues meet once, in front of the copy back.  (The exact moments are
    // VALU-bound, the batch streams L^-1.)  GPMPC_ROLLOUT_OVERLAP=0: one group after the other, step by step, on the main queue.
    static const bool overlap_env = !(getenv("GPMPC_ROLLOUT_OVERLAP") && atoi(getenv("GPMPC_ROLLOUT_OVERLAP")) == 0);
    // (only 'EM' has scratch of its own; the legacy methods form their cross-covariances in the buffers the 'ME' / 'TA' batch uses)
    const bool overlap = overlap_env && moments && nA > 0 && !cnt[GPMPC_OLD_ME] && !cnt[GPMPC_OLD_TA] && h->side_stream &&
                         h->stream == h->own_stream;
    hipStream_t main_q = h->stream;
    int rc = GPMPC_OK;
    auto moments_prologue = [&]() -> int {       // K^-1 and beta, on the handle's current queue
        if (!h->have_invK) {
            PhaseTimer t(h, GPMPC_PH_INVK);
            CHK(compute_invK(h->cx(), h->ws));
            h->have_invK = true;
        }
        return ensure_beta(h);
    };
    auto feed = [&](int t, int off, int cntg) {  // (mean, cov)_{t-1} and u_t of trajectories [off, off + cntg) -> their next inputs
        const double* pM = dM + (size_t)(t - 1) * M * Ny;
        const double* pC = dC + (size_t)(t - 1) * M * Ny * Ny;
        hipLaunchKernelGGL(rollout_feed_multi_kernel, dim3(cntg), dim3(64), 0, h->stream, pM + (size_t)off * Ny, pC + (size_t)off * Ny * Ny,
                           dU + ((size_t)t * M + off) * nu1, dsa, dsb, dZ + (size_t)off * d, dS + (size_t)off * d * d, Ny, d, nu1);
    };
    auto step_a = [&](int t) -> int {            // the 'ME' / 'TA' batch of time step t
        double* oM = dM + (size_t)t * M * Ny;
        double* oC = dC + (size_t)t * M * Ny * Ny;
        CHK(predict_chunk(h, nA, dZ, oM, dV, nTA ? dJ : nullptr));
        if (nME)
            hipLaunchKernelGGL(cov_assemble_kernel, dim3((unsigned)(((long)nME * Ny * Ny + 255) / 256)), dim3(256), 0, h->stream, dV, dJ,
                               (const double*)nullptr, oC, nME, Ny, d);
        if (nTA)
            hipLaunchKernelGGL(cov_assemble_kernel, dim3((unsigned)(((long)nTA * Ny * Ny + 255) / 256)), dim3(256), 0, h->stream,
                               dV + (size_t)nME * Ny, dJ + (size_t)nME * Ny * d, dS + (size_t)nME * d * d, oC + (size_t)nME * Ny * Ny,
                               nTA, Ny, d);
        return GPMPC_OK;
    };
    auto step_m = [&](int t) -> int {            // the moment-method trajectories of time step t
        double* oM = dM + (size_t)t * M * Ny;
        double* oC = dC + (size_t)t * M * Ny * Ny;
        int off = nA;
        for (int code : {GPMPC_EM, GPMPC_OLD_ME, GPMPC_OLD_TA}) {
            if (!cnt[code]) continue;
            CHK(predict_moments_chunk(h, code, cnt[code], dZ + (size_t)off * d, dS + (size_t)off * d * d, oM + (size_t)off * Ny,
                                      oC + (size_t)off * Ny * Ny));
            off += cnt[code];
        }
        return GPMPC_OK;
    };
    if (overlap) {
        alpha_ready(h);                          // (the workers' queue may still carry a fit's alpha: the main queue is ordered behind it)
        hipEventRecord(h->ev_fork, main_q);      // the inputs' upload, and whatever produced the factors
        hipStreamWaitEvent(h->side_stream, h->ev_fork, 0);
        h->stream = h->side_stream;              // (everything below enqueues on the handle's current queue)
        rc = moments_prologue();
        for (int t = 0; t < T && rc == GPMPC_OK; ++t) {
            if (t > 0) feed(t, nA, M - nA);
            rc = step_m(t);
        }
        hipEventRecord(h->ev_join, h->side_stream);
        h->stream = main_q;
        for (int t = 0; t < T && rc == GPMPC_OK; ++t) {
            if (t > 0) feed(t, 0, nA);
            rc = step_a(t);
        }
        hipStreamWaitEvent(main_q, h->ev_join, 0);
    } else {
        if (moments) rc = moments_prologue();
        for (int t = 0; t < T && rc == GPMPC_OK; ++t) {
            if (t > 0) feed(t, 0, M);
            if (nA > 0) rc = step_a(t);
            if (rc == GPMPC_OK && moments) rc = step_m(t);
        }
    }
    if (rc != GPMPC_OK) { hipStreamSynchronize(h->stream); if (h->side_stream) hipStreamSynchronize(h->side_stream); return rc; }
    hipError_t e = hipMemcpyAsync(h->rollm_pin + (dM - buf), dM, (nM + nC) * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail(GPMPC_EHIP, "%s", hipGetErrorString(e));
    const double* hM = h->rollm_pin + (dM - buf);
    const double* hC = h->rollm_pin + (dC - buf);
    for (int k = 0; k < M; ++k)
        for (int t = 0; t < T; ++t) {
            std::memcpy(mean + ((size_t)perm[k] * T + t) * Ny, hM + ((size_t)t * M + k) * Ny, Ny * sizeof(double));
            std::memcpy(cov + ((size_t)perm[k] * T + t) * Ny * Ny, hC + ((size_t)t * M + k) * Ny * Ny, (size_t)Ny * Ny * sizeof(double));
        }
    return GPMPC_OK;
}
