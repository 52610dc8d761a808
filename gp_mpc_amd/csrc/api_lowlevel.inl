// api_lowlevel.inl -- part of gpmpc_api.hip (one translation unit; included in order, not compiled alone).
// Concern: low-level dense ops for the parity tests, tuning switches.
// ------------------------------------------------------------------------------------------------
// low-level dense ops for the parity tests
// ------------------------------------------------------------------------------------------------
extern "C" int gpmpc_cholesky(int device, int n, double* A, double* Ainv, int* info) {
    if (n <= 0 || !A || !info) return fail(GPMPC_EINVAL, "bad arguments");
    CHK(ensure_device(device));
    const int Np = round_up(n, 64);
    Workspace ws;
    CHK(ws_alloc(ws, 1, Np, 1));
    std::vector<double> tmp((size_t)Np * Np, 0.0);
    for (int i = 0; i < n; ++i) std::memcpy(tmp.data() + (size_t)i * Np, A + (size_t)i * n, n * sizeof(double));
    for (int i = n; i < Np; ++i) tmp[(size_t)i * Np + i] = 1.0;
    HIPCHK(hipMemcpy(ws.K, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    Ctx cx{nullptr, g_crow_mode[device]};
    HIPCHK(hipMemset(ws.info, 0, sizeof(int)));
    factor_blocked(cx, ws, true);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(info, ws.info, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tmp.data(), ws.L, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) std::memcpy(A + (size_t)i * n, tmp.data() + (size_t)i * Np, n * sizeof(double));
    if (Ainv) {
        HIPCHK(hipMemcpy(tmp.data(), ws.Inv, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) std::memcpy(Ainv + (size_t)i * n, tmp.data() + (size_t)i * Np, n * sizeof(double));
    }
    ws_free(ws);
    return GPMPC_OK;
}

extern "C" int gpmpc_dgemm(int device, int transa, int transb, int M, int N, int K, double alpha, const double* A,
                           int lda, const double* B, int ldb, double beta, double* C, int ldc) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return fail(GPMPC_EINVAL, "bad arguments");
    CHK(ensure_device(device));
    // repack into padded device buffers: K padded to 16, leading dimensions even
    const int Kp = round_up(K, 16), Mp = round_up(M, 2), Nq = round_up(N, 2);
    const int rowsA = transa ? Kp : M, colsA = transa ? Mp : Kp;
    const int rowsB = transb ? N : Kp, colsB = transb ? Kp : Nq;
    std::vector<double> a((size_t)rowsA * colsA, 0.0), b((size_t)rowsB * colsB, 0.0);
    for (int i = 0; i < (transa ? K : M); ++i)
        std::memcpy(a.data() + (size_t)i * colsA, A + (size_t)i * lda, (transa ? M : K) * sizeof(double));
    for (int i = 0; i < (transb ? N : K); ++i)
        std::memcpy(b.data() + (size_t)i * colsB, B + (size_t)i * ldb, (transb ? K : N) * sizeof(double));
    double *dA, *dB, *dC;
    HIPCHK(hipMalloc(&dA, a.size() * sizeof(double)));
    HIPCHK(hipMalloc(&dB, b.size() * sizeof(double)));
    HIPCHK(hipMalloc(&dC, (size_t)M * N * sizeof(double)));
    HIPCHK(hipMemcpy(dA, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dB, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> c((size_t)M * N);
    for (int i = 0; i < M; ++i) std::memcpy(c.data() + (size_t)i * N, C + (size_t)i * ldc, N * sizeof(double));
    HIPCHK(hipMemcpy(dC, c.data(), c.size() * sizeof(double), hipMemcpyHostToDevice));
    Ctx cx{nullptr, g_crow_mode[device]};
    GemmP p = gemm_base(cx);
    p.A = dA; p.lda = colsA; p.a_mc = transa ? 1 : 0;
    p.B = dB; p.ldb = colsB; p.b_nc = transb ? 0 : 1;
    p.C = dC; p.ldc = N;
    p.M = M; p.N = N; p.K = Kp; p.alpha = alpha; p.beta = beta;
    // GPMPC_DGEMM_TILE=128|64|32 pins the tile (tests reach the large-tile kernels with small matrices)
    launch_gemm(p, 1, cx.stream, getenv("GPMPC_DGEMM_TILE") ? atoi(getenv("GPMPC_DGEMM_TILE")) : 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(c.data(), dC, c.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < M; ++i) std::memcpy(C + (size_t)i * ldc, c.data() + (size_t)i * N, N * sizeof(double));
    hipFree(dA); hipFree(dB); hipFree(dC);
    return GPMPC_OK;
}

extern "C" int gpmpc_schedule_stats(int mode, int tilesM, int tilesN, int batch, int K, int slots, double* stats) {
    if (!stats || (mode != PG_VAR && mode != PG_XTX) || tilesM <= 0 || tilesN <= 0 || batch <= 0 || batch >= 256 || slots <= 0 ||
        tilesM >= 4096 || tilesN >= 4096 || K <= 0)
        return fail(GPMPC_EINVAL, "bad arguments");
    const VarSchedule s = mode == PG_VAR ? var_schedule(tilesM, tilesN, batch, K, slots) : xtx_schedule(tilesM, batch, K, slots);
    std::map<int, int> seen;
    for (int w : s.list) ++seen[w];
    long expected = 0, wrong = 0;
    for (int z = 0; z < batch; ++z)
        for (int tm = 0; tm < tilesM; ++tm)
            for (int tn = 0; tn < (mode == PG_VAR ? tilesN : tm + 1); ++tn) {
                ++expected;
                auto it = seen.find(var_tile_word(z, tm, tn));
                if (it == seen.end() || it->second != 1) ++wrong;
            }
    if ((long)s.list.size() != expected) wrong += std::labs((long)s.list.size() - expected);
    int longest = 0;
    for (int i = 0; i < slots; ++i) longest = std::max(longest, s.off[i + 1] - s.off[i]);
    stats[0] = (double)s.list.size(); stats[1] = s.max_load; stats[2] = s.mean_load; stats[3] = s.home; stats[4] = (double)wrong;
    stats[5] = longest;
    return GPMPC_OK;
}

extern "C" int gpmpc_set_tuning(const char* name, int value) {
    if (!name) return fail(GPMPC_EINVAL, "NULL name");
    if (std::strcmp(name, "gemm_tile") == 0) {
        if (value != 0 && value != 32 && value != 64 && value != 128) return fail(GPMPC_EINVAL, "gemm_tile must be 0, 32, 64 or 128");
        g_gemm_force_tile = value;
        return GPMPC_OK;
    }
    if (std::strcmp(name, "cu_count") == 0) {           // pretend device 0 has fewer compute units (worker counts follow)
        CHK(ensure_device(0));
        static int real = g_cu_count[0];
#ifndef GPMPC_EMULATED
        if (value < 8 || value > real) return fail(GPMPC_EINVAL, "cu_count must be in [8, %d]", real);
#else
        if (value < 8 || value > 64) return fail(GPMPC_EINVAL, "cu_count must be in [8, 64]");
        (void)real;
#endif
        g_cu_count[0] = value;
        return GPMPC_OK;
    }
    if (std::strcmp(name, "worker_courier") == 0) {      // which worker kernel the chained factorisation launches (chol_worker.hpp)
        if (value < -1 || value > 1) return fail(GPMPC_EINVAL, "worker_courier must be -1 (default), 0 or 1");
        g_worker_courier = value;
        return GPMPC_OK;
    }
    if (std::strcmp(name, "handoff_write_through") == 0) {   // hand-offs inside the chained factorisation: write-through stores + drained flag / release fence
        if (value < -1 || value > 1) return fail(GPMPC_EINVAL, "handoff_write_through must be -1 (default), 0 or 1");
        g_handoff_wt = value;
        return GPMPC_OK;
    }
    if (std::strcmp(name, "vargemm_persist") == 0) {     // variance product: 0 one tile per workgroup, 1 persistent static schedule, 2 ... at any size
        if (value < -1 || value > 2) return fail(GPMPC_EINVAL, "vargemm_persist must be -1 (default), 0, 1 or 2");
        g_vargemm_persist = value;
        return GPMPC_OK;
    }
    if (std::strcmp(name, "train_batch_cap") == 0) {     // lock-step restart search: points per batch (0 = automatic, 1 = one at a time)
        if (value < 0 || value > TRAIN_BATCH_CAP_MAX) return fail(GPMPC_EINVAL, "train_batch_cap must be in [0, %d]", TRAIN_BATCH_CAP_MAX);
        g_train_batch_cap = value;
        return GPMPC_OK;
    }
    if (std::strcmp(name, "em_chunk") == 0) {            // exact-moment pair sums: column tiles per workgroup (0 = default, em_kernels.hpp)
        if (value < 0) return fail(GPMPC_EINVAL, "em_chunk must be >= 0");
        g_em_chunk = value;
        return GPMPC_OK;
    }
    if (std::strcmp(name, "em_diag_segs") == 0) {        // exact-moment a == b pair sums: ranges per pair of the balanced schedule (-1 = default, 0 = strips and chunks)
        if (value < -1) return fail(GPMPC_EINVAL, "em_diag_segs must be >= -1");
        g_em_diag_segs = value;
        return GPMPC_OK;
    }
    if (std::strcmp(name, "fail_nll_after") == 0) {      // fault injection for the tests of the restart shard's failure paths
        if (value < 0) return fail(GPMPC_EINVAL, "fail_nll_after must be >= 0");
        static const bool testing = getenv("GPMPC_TESTING") && atoi(getenv("GPMPC_TESTING")) != 0;
        if (!testing) return fail(GPMPC_EINVAL, "fail_nll_after is a test knob: start the process with GPMPC_TESTING=1");
        g_fail_nll_after.store(value);
        return GPMPC_OK;
    }
    return fail(GPMPC_EINVAL, "unknown tuning knob '%s'", name);
}

extern "C" int gpmpc_kernel_matrix(int device, int n1, int n2, int d, const double* X, const double* Z, const double* ell,
                                   double sf2, double* out) {
    if (n1 <= 0 || n2 <= 0 || d <= 0 || !X || !Z || !ell || !out) return fail(GPMPC_EINVAL, "bad arguments");
    CHK(ensure_device(device));
    double *dX, *dZ, *dE, *dO;
    HIPCHK(hipMalloc(&dX, (size_t)n1 * d * sizeof(double)));
    HIPCHK(hipMalloc(&dZ, (size_t)n2 * d * sizeof(double)));
    HIPCHK(hipMalloc(&dE, (size_t)d * sizeof(double)));
    HIPCHK(hipMalloc(&dO, (size_t)n1 * n2 * sizeof(double)));
    HIPCHK(hipMemcpy(dX, X, (size_t)n1 * d * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dZ, Z, (size_t)n2 * d * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dE, ell, (size_t)d * sizeof(double), hipMemcpyHostToDevice));
    const long ne = (long)n1 * n2;
    hipLaunchKernelGGL(kernel_matrix_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, 0, dX, dZ, dE, sf2, dO, n1, n2, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dO, (size_t)ne * sizeof(double), hipMemcpyDeviceToHost));
    hipFree(dX); hipFree(dZ); hipFree(dE); hipFree(dO);
    return GPMPC_OK;
}

