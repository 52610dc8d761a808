// Moment-based methods: 'EM' (a11) and the legacy 'old_ME' / 'old_TA' (a12).  Included by gpmpc_api.hip.
static int g_em_chunk = 0;          // gpmpc_set_tuning("em_chunk", n): column tiles per workgroup of the pair sums (0 = default)
static int g_em_diag_segs = -1;     // gpmpc_set_tuning("em_diag_segs", n): ranges per a == b pair (-1 = default, 0 = strips and chunks)

// Ranges per a == b pair of the balanced schedule (em_diag_kernel) when nobody asks for a number: six workgroups per CU over
// the outputs (four are resident; the a != b launch runs next to them), ranges of at least 24 tiles as long as that still
// leaves one workgroup per CU.  C3 (N = 8192, Ny = 6, 256 CUs; EM roll-out of 30 steps, profiles/r06_sweep_em_diag_segs.txt):
// 128 / 170 / 256 / 344 / 512 / 768 / 1032 / 2064 ranges -> 30.7 / 32.4 / 30.6 / 31.4 / 31.7 / 32.2 / 33.3 / 36.1 ms
// (strips and chunks: 35.3); the rule gives 256 there.
static int em_diag_default_segs(int cus, int Ny, long tri) {
    const long fill = std::max(1, 6 * cus / Ny), one_per_cu = std::max(1, cus / Ny);
    return (int)std::max<long>(1, std::min(fill, std::max(one_per_cu, tri / 24)));
}

// beta_a = K_a^-1 y_a (gp_functions.py:383; the reference multiplies the explicit inverse)
static int ensure_beta(gpmpc_gp* h) {
    if (h->have_beta) return GPMPC_OK;
    if (!h->beta) HIPCHK(hipMalloc(&h->beta, (size_t)h->Ny * h->Np * sizeof(double)));
    const Ctx cx = h->cx();
    hipLaunchKernelGGL(gemv_rows_kernel, dim3(h->Np / 4, h->Ny), dim3(256), 0, cx.stream, h->ws.InvK, h->Y, h->beta, h->Np,
                       h->ws.mat(), (long)h->Np, (long)h->Np, 0);
    // padded rows of K^-1 are identity rows against a zero-padded y: beta stays 0 there
    h->have_beta = true;
    return GPMPC_OK;
}

static int ensure_em_scratch(gpmpc_gp* h, long bytes, bool sens = false) {
    double*& buf = sens ? h->ems : h->em;
    long& have = sens ? h->emsBytes : h->emBytes;
    if (bytes <= have) return GPMPC_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    hipFree(buf);
    buf = nullptr;
    have = 0;
    HIPCHK(hipMalloc(&buf, (size_t)bytes));
    have = bytes;
    return GPMPC_OK;
}

static int predict_moments_chunk(gpmpc_gp* h, int method, int B, const double* dZ, const double* dSigma,
                                 double* dMean, double* dCov) {
    const Ctx cx = h->cx();
    const int Np = h->Np, Ny = h->Ny, d = h->d, N = h->N;
    CHK(ensure_beta(h));
    if (method == GPMPC_EM) {
        PhaseTimer t(h, GPMPC_PH_EM);
        const int KD = em_depth(d);                           // 8 or 16 (d <= DMAX = 16 is checked at gpmpc_create)
        const int P = Ny * (Ny + 1) / 2, tiles = Np / 64;
        // column tiles per workgroup of the pair sums (em_kernels.hpp; GPMPC_EM_CHUNK, tuning aid; >= tiles: one workgroup per strip)
        // (also gpmpc_set_tuning("em_chunk", n): the tests sweep it at small sizes)
        static const int em_chunk_env = getenv("GPMPC_EM_CHUNK") ? atoi(getenv("GPMPC_EM_CHUNK")) : 64;   // (C3: 64 -> 38.4 ms, 32 -> 39.4, 16 -> 42.5, whole strips 39.9: profiles/r05_em_chunk_ab.txt)
        const int em_chunk = std::max(1, std::min(g_em_chunk > 0 ? g_em_chunk : em_chunk_env, tiles)), nstrip = tiles * ((tiles + em_chunk - 1) / em_chunk);
        // The a == b launch on a balanced schedule (em_diag_kernel): the pair's triangle of tiles in `diag_segs` equal ranges.
        // GPMPC_EM_DIAG_SEGS / gpmpc_set_tuning("em_diag_segs", n): 0 = strips and chunks as r05 (em_pair2_kernel<true>), n > 0 =
        // that many ranges per pair.  nslots = partial sums per pair (both launches write all of them).
        static const int diag_segs_env = getenv("GPMPC_EM_DIAG_SEGS") ? atoi(getenv("GPMPC_EM_DIAG_SEGS")) : -1;
        const long tri = (long)tiles * (tiles + 1) / 2;
        const int diag_want = g_em_diag_segs >= 0 ? g_em_diag_segs : diag_segs_env >= 0 ? diag_segs_env : em_diag_default_segs(g_cu_count[h->device], Ny, tri);
        const int diag_segs = (int)std::min<long>(std::min(diag_want, 4096), tri);
        const int nslots = std::max(nstrip, diag_segs);
        const long prepN = (long)B * (Ny + P) * (d * d + 1), partN = (long)B * P * nslots;
        const long opsN = (long)B * P * (2 * KD + 2) * Np, mpartN = (long)B * Ny * EM_MEAN_CHUNKS, bndN = (long)B * P * 4;
        CHK(ensure_em_scratch(h, (prepN + partN + opsN + mpartN + bndN) * (long)sizeof(double)));
        double* prep = h->em;
        double* partial = h->em + prepN;
        double* ops = partial + partN;
        double* mpart = ops + opsN;
        unsigned long long* bnd = reinterpret_cast<unsigned long long*>(mpart + mpartN);   // operand magnitudes per (input, pair)
        hipLaunchKernelGGL(em_prep_kernel, dim3((unsigned)(B * (Ny + P))), dim3(DMAX * GJ_LD), 0, cx.stream, h->ws.hyper, dSigma,
                           prep, B, Ny, d, bnd);
        // The two pair-sum launches are independent (a != b pairs / a == b pairs, disjoint partial sums): at sizes beyond the
        // captured-graph range the a == b launch -- the one that streams K^-1 -- goes to the inverse queue NEXT TO the other
        // (GPMPC_EM_PAIR_OVERLAP=0: one after the other, as r01-r05), and in front of it, next to the operands kernel, the mean
        // (it needs the prepared matrices only; the covariance's last kernel waits for that queue anyway).
        static const bool pair_overlap_env = !(getenv("GPMPC_EM_PAIR_OVERLAP") && atoi(getenv("GPMPC_EM_PAIR_OVERLAP")) == 0);
        const bool pair_overlap = pair_overlap_env && dCov && h->aux_stream && Np > 2048 && Ny > 1;
        hipStream_t diag_q = pair_overlap ? h->aux_stream : cx.stream;
        if (pair_overlap) {
            hipEventRecord(TailState::get(h->tail.ev_tail), cx.stream);
            hipStreamWaitEvent(diag_q, h->tail.ev_tail, 0);
        }
        hipLaunchKernelGGL(em_mean_kernel, dim3(Ny, B, EM_MEAN_CHUNKS), dim3(256), 0, diag_q, h->XT, dZ, h->beta, prep, mpart,
                           N, Np, d, Ny);
        if (!dCov) {     // mean only (gpmpc_predict_em_sens without the covariance value)
            hipLaunchKernelGGL(em_mean_finish_kernel, dim3((B * Ny + 255) / 256), dim3(256), 0, diag_q, mpart, dMean, B * Ny);
            HIPCHK(hipGetLastError());
            return GPMPC_OK;
        }
        // (with the covariance the chunk sums of the mean are added up by em_finish_kernel)
        // exp of the pair sums: GPMPC_EM_PAIR (tuning aid) 1 (default) the 2^(j / 2048) table, 2 the conflict-free 32-entry table,
        // 3 the polynomial exp_lean (r05, C3, same box: 40.8 / 43.3 / 46.3 ms per step; r04's kernel 45.7)
        static const int pair_form = getenv("GPMPC_EM_PAIR") ? atoi(getenv("GPMPC_EM_PAIR")) : 1;
        const double* etab = g_exp_tab[h->device];
#define GPMPC_EM_PAIR2(KDV, TABV)                                                                                                     \
        if (pair_overlap) {                                                                                                           \
            hipEventRecord(TailState::get(h->tail.ev_ks), cx.stream);                                                                 \
            hipStreamWaitEvent(diag_q, h->tail.ev_ks, 0);                                                                             \
        }                                                                                                                             \
        /* each launch twice: without / with the table exp's clamp; the workgroups of the one a pair does not need leave at once */  \
        if (diag_segs > 0) {                                                                                                          \
            hipLaunchKernelGGL((em_diag_kernel<KDV, TABV, false>), dim3(nslots, Ny, B), dim3(256), 0, diag_q, ops, h->beta,           \
                               h->ws.InvK, partial, N, Np, Ny, cx.crow_mode, etab, diag_segs, nslots, bnd);                           \
            hipLaunchKernelGGL((em_diag_kernel<KDV, TABV, true>), dim3(nslots, Ny, B), dim3(256), 0, diag_q, ops, h->beta,            \
                               h->ws.InvK, partial, N, Np, Ny, cx.crow_mode, etab, diag_segs, nslots, bnd);                           \
        } else {                                                                                                                      \
            hipLaunchKernelGGL((em_pair2_kernel<true, KDV, TABV, false>), dim3(nslots, P, B), dim3(256), 0, diag_q, ops, h->beta,     \
                               h->ws.InvK, partial, N, Np, Ny, cx.crow_mode, etab, em_chunk, nslots, bnd);                            \
            hipLaunchKernelGGL((em_pair2_kernel<true, KDV, TABV, true>), dim3(nslots, P, B), dim3(256), 0, diag_q, ops, h->beta,      \
                               h->ws.InvK, partial, N, Np, Ny, cx.crow_mode, etab, em_chunk, nslots, bnd);                            \
        }                                                                                                                             \
        if (pair_overlap) hipEventRecord(TailState::get(h->tail.ev_mean), diag_q);                                                    \
        hipLaunchKernelGGL((em_pair2_kernel<false, KDV, TABV, false>), dim3(nslots, P, B), dim3(256), 0, cx.stream, ops, h->beta,     \
                           h->ws.InvK, partial, N, Np, Ny, cx.crow_mode, etab, em_chunk, nslots, bnd);                                \
        hipLaunchKernelGGL((em_pair2_kernel<false, KDV, TABV, true>), dim3(nslots, P, B), dim3(256), 0, cx.stream, ops, h->beta,      \
                           h->ws.InvK, partial, N, Np, Ny, cx.crow_mode, etab, em_chunk, nslots, bnd);                                \
        if (pair_overlap) hipStreamWaitEvent(cx.stream, h->tail.ev_mean, 0);
#define GPMPC_EM_PAIR2_ANY(KDV)                                                                                                       \
        if (pair_form == 2) { GPMPC_EM_PAIR2(KDV, 2) } else if (pair_form == 3) { GPMPC_EM_PAIR2(KDV, 0) } else { GPMPC_EM_PAIR2(KDV, 1) }
        if (KD == 8) {
            hipLaunchKernelGGL((em_operands_kernel<8>), dim3((Np + 255) / 256, P, B), dim3(256), 0, cx.stream, h->XT, dZ, h->ws.hyper,
                               prep, ops, N, Np, d, Ny, bnd);
            GPMPC_EM_PAIR2_ANY(8)
        } else {      // d = 9 .. 16: the same kernels with a 16-deep cross term (gp_exact_moment is dimension-generic)
            hipLaunchKernelGGL((em_operands_kernel<16>), dim3((Np + 255) / 256, P, B), dim3(256), 0, cx.stream, h->XT, dZ, h->ws.hyper,
                               prep, ops, N, Np, d, Ny, bnd);
            GPMPC_EM_PAIR2_ANY(16)
        }
#undef GPMPC_EM_PAIR2
#undef GPMPC_EM_PAIR2_ANY
        hipLaunchKernelGGL(em_finish_kernel, dim3((unsigned)((long)B * P)), dim3(64), 0, cx.stream, partial, prep,
                           h->ws.hyper, dMean, dCov, B, Ny, d, nslots, mpart);
        HIPCHK(hipGetLastError());
        return GPMPC_OK;
    }
    // legacy: u = K^-1 ks for the whole batch as one GEMM  UT[j][:] = KsT[j][:] K^-1
    const int Bp = round_up(B, 64);
    const size_t utBytes = (size_t)Ny * h->Bcap * Np * sizeof(double);
    if (!h->UT) HIPCHK(hipMalloc(&h->UT, utBytes));
    CHK(ensure_em_scratch(h, (long)B * Ny * 4 * (long)sizeof(double)));
    {
        PhaseTimer t(h, GPMPC_PH_CROSSCOV);
        launch_crosscov(cx.stream, d, h->XT, h->ws.hyper, h->ws.alpha, dZ, h->KsT, h->meanT, nullptr, N, Np, B, Bp, Ny);
    }
    PhaseTimer t(h, GPMPC_PH_EM);
    GemmP p = gemm_base(cx);
    p.A = h->KsT; p.lda = Np; p.sA = (long)Bp * Np; p.a_mc = 0;
    p.B = h->ws.InvK; p.ldb = Np; p.sB = h->ws.mat(); p.b_nc = 1;
    p.C = h->UT; p.ldc = Np; p.sC = (long)Bp * Np;
    p.M = Bp; p.N = Np; p.K = Np;
    launch_gemm(p, Ny, cx.stream);
    hipLaunchKernelGGL(legacy_scalars_kernel, dim3(B, Ny), dim3(256), 0, cx.stream, h->XT, dZ, h->ws.hyper, h->Y, h->beta,
                       h->KsT, h->UT, h->em, N, Np, d, Bp, Ny);
    hipLaunchKernelGGL(legacy_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, cx.stream, h->em, h->XT, dZ, h->ws.hyper,
                       dSigma, dMean, dCov, B, Ny, d, method == GPMPC_OLD_TA ? 1 : 0);
    HIPCHK(hipGetLastError());
    return GPMPC_OK;
}
