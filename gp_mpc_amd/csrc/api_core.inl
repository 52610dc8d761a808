// api_core.inl -- part of gpmpc_api.hip (one translation unit; included in order, not compiled alone).
// Concern: errors, device bring-up + fp64 MFMA self-test, device block list, factorisation workspace, profile brackets, launch context.
// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(GPMPC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                        __LINE__);                                                                    \
    } while (0)
#define CHK(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != GPMPC_OK) return rc_; \
    } while (0)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------
// device bring-up + fp64 MFMA self-test
// ------------------------------------------------------------------------------------------------
// The persistent-kernel factorisation wants the whole chip (one CU-filling worker per CU and a CU for the chain): two of
// them at once starve each other's workgroups of the residency their hand-offs rely on.  Handles of one process
// therefore take turns on the device (a factorisation is ~2 ms at N = 4096).
static std::mutex g_factor_mutex[64];
static int g_crow_mode[64];
static int g_cu_count[64];
static double* g_exp_tab[64];          // per device: the table of exp_tab (gp_kernels.hpp)
static bool g_dev_ready[64];

static int mfma_selftest(int device, int* layout_out, double* tflops_out) {
    HIPCHK(hipSetDevice(device));
    double hA[64], hB[64], hD[256];
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1.0 + i * 0.25 - k * 0.5 + 0.03125 * i * k;
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 16; ++j) hB[k * 16 + j] = -2.0 + 0.5 * j + 0.125 * k * k - 0.0625 * j * k;
    double *dA, *dB, *dD;
    HIPCHK(hipMalloc(&dA, sizeof(hA)));
    HIPCHK(hipMalloc(&dB, sizeof(hB)));
    HIPCHK(hipMalloc(&dD, sizeof(hD)));
    HIPCHK(hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost));
    int layout = -1;
    for (int mode = 0; mode < 2 && layout < 0; ++mode) {
        bool ok = true;
        for (int l = 0; l < 64 && ok; ++l)
            for (int r = 0; r < 4 && ok; ++r) {
                const int row = mode == 0 ? (l >> 4) + 4 * r : 4 * (l >> 4) + r, col = l & 15;
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s += hA[row * 4 + k] * hB[k * 16 + col];
                if (std::fabs(s - hD[l * 4 + r]) > 1e-12 * (1.0 + std::fabs(s))) ok = false;
            }
        if (ok) layout = mode;
    }
    if (layout_out) *layout_out = layout;
    if (tflops_out) {
        *tflops_out = 0.0;
#ifndef GPMPC_EMULATED
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        // 4 workgroups x 4 waves per CU = 4 waves per SIMD (one wave alone can only issue an f64 MFMA every
        // ~142 cycles); long enough that the ramp and tail of the launch do not matter
        const int blocks = prop.multiProcessorCount * 4, iters = 4096;   // ~1 ms
        double* dOut;
        HIPCHK(hipMalloc(&dOut, (size_t)blocks * 256 * sizeof(double)));
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(mfma_rate_kernel, dim3(blocks), dim3(256), 0, 0, dOut, 64);
        HIPCHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(mfma_rate_kernel, dim3(blocks), dim3(256), 0, 0, dOut, iters);
        HIPCHK(hipEventRecord(e1, 0));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)blocks * 4 /*waves*/ * iters * 4.0 * 2.0 * 16 * 16 * 4;
        *tflops_out = flops / (ms * 1e-3) * 1e-12;
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        hipFree(dOut);
#endif
    }
    hipFree(dA);
    hipFree(dB);
    hipFree(dD);
    if (layout < 0)
        return fail(GPMPC_EHIP, "v_mfma_f64_16x16x4_f64 returned a fragment layout this library does not know");
    return GPMPC_OK;
}

static int ensure_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(GPMPC_EHIP, "no HIP device visible (libgpmpc_hip needs an MI355X / gfx950 GPU)");
    if (device < 0 || device >= n || device >= 64) return fail(GPMPC_EINVAL, "device %d out of range (count %d)", device, n);
    HIPCHK(hipSetDevice(device));
    if (!g_dev_ready[device]) {
#ifndef GPMPC_EMULATED
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(GPMPC_EHIP, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        g_cu_count[device] = prop.multiProcessorCount;
#else
        g_cu_count[device] = getenv("GPMPC_EMU_CUS") ? atoi(getenv("GPMPC_EMU_CUS")) : 8;
#endif
        if (!g_exp_tab[device]) {                               // 2^(j / 2048), correctly rounded (exp_tab, gp_kernels.hpp)
            std::vector<double> tab(EXPT_N);
            for (int j = 0; j < EXPT_N; ++j) tab[j] = (double)exp2l((long double)j / (long double)EXPT_N);
            HIPCHK(hipMalloc(&g_exp_tab[device], EXPT_N * sizeof(double)));
            HIPCHK(hipMemcpy(g_exp_tab[device], tab.data(), EXPT_N * sizeof(double), hipMemcpyHostToDevice));
        }
        int layout = -1;
        CHK(mfma_selftest(device, &layout, nullptr));
        g_crow_mode[device] = layout;
        g_dev_ready[device] = true;
    }
    return GPMPC_OK;
}

// ------------------------------------------------------------------------------------------------
// factorisation workspace: K (destroyed), L, L^-1, scratch, w, alpha for `batch` matrices
// ------------------------------------------------------------------------------------------------
// rows per segment of the pipelined triangular inverse (power of two times 64); small in the emulated
// build so that the CPU tests reach the pipelined path at N ~ 600
#ifdef GPMPC_EMULATED
static const int SEGR = 128;
#else
static const int SEGR = 512;
#endif

// Large device blocks (the N x N matrices of a workspace) come from size classes -- a quarter of the power of two
// below the request -- and go back to a small per-process list instead of to the driver: gpmpc_append builds its new
// workspace before it drops the old one, and a fresh multi-GB hipMalloc was measured at anything between 0.3 ms and
// 0.5 s on the same box (append +64 at C3 size: 9 ms or 500 ms).  With classes the blocks the previous append gave back fit
// the next one (8-9 appends of 64 points per class at N = 8192).  The list is emptied when the last handle goes.
struct DevBlock { void* p; size_t cls; int dev; };
static std::mutex g_block_mutex;
static std::vector<DevBlock> g_free_blocks, g_live_blocks;
static int g_live_handles = 0;
static long g_block_reuses = 0, g_block_fresh = 0;   // process-wide, read through gpmpc_get_counter
constexpr size_t BLOCK_MIN = (size_t)64 << 20;
constexpr size_t BLOCK_LIST_MAX = 16;

static size_t block_class(size_t bytes) {
    size_t p2 = 1;
    while (p2 * 2 <= bytes) p2 *= 2;
    const size_t g = p2 / 4;
    return (bytes + g - 1) / g * g;
}

static void block_list_release();

static hipError_t block_alloc(double** out, size_t bytes) {
    if (bytes < BLOCK_MIN) return hipMalloc(out, bytes);
    const size_t cls = block_class(bytes);
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_block_mutex);
        for (size_t i = 0; i < g_free_blocks.size(); ++i)
            if (g_free_blocks[i].cls == cls && g_free_blocks[i].dev == dev) {
                *out = (double*)g_free_blocks[i].p;
                ++g_block_reuses;
                g_live_blocks.push_back(g_free_blocks[i]);
                g_free_blocks.erase(g_free_blocks.begin() + i);
                return hipSuccess;
            }
    }
    hipError_t e = hipMalloc(out, cls);
    size_t got = cls;
    if (e != hipSuccess) {                 // out of memory with the class rounding: give the idle blocks back, then ask for the exact size
        (void)hipGetLastError();
        block_list_release();
        e = hipMalloc(out, cls);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            got = bytes;
            e = hipMalloc(out, bytes);
        }
    }
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_block_mutex);
        ++g_block_fresh;
        g_live_blocks.push_back({(void*)*out, got, dev});
    }
    return e;
}

// bytes the idle list holds on `dev` (handed out again before any fresh allocation: they count as free for sizing decisions)
static size_t block_list_idle_bytes(int dev) {
    std::lock_guard<std::mutex> lk(g_block_mutex);
    size_t s = 0;
    for (const DevBlock& b : g_free_blocks)
        if (b.dev == dev) s += b.cls;
    return s;
}

static void block_free(double* p) {
    if (!p) return;
    (void)hipDeviceSynchronize();      // what hipFree implies: nothing in flight may still touch a block that is handed out again
    {
        std::lock_guard<std::mutex> lk(g_block_mutex);
        for (size_t i = 0; i < g_live_blocks.size(); ++i)
            if (g_live_blocks[i].p == (void*)p) {
                const DevBlock b = g_live_blocks[i];
                g_live_blocks.erase(g_live_blocks.begin() + i);
                if (g_free_blocks.size() < BLOCK_LIST_MAX) {
                    g_free_blocks.push_back(b);
                    return;
                }
                break;
            }
    }
    hipFree(p);
}

static void block_list_release() {
    std::vector<DevBlock> drop;
    {
        std::lock_guard<std::mutex> lk(g_block_mutex);
        drop.swap(g_free_blocks);
    }
    for (auto& b : drop) hipFree(b.p);
}

struct Workspace {
    int batch = 0, Np = 0, d = 0;
    double *K = nullptr, *L = nullptr, *Inv = nullptr, *InvK = nullptr, *W = nullptr;
    double* Wl = nullptr;   // level scratch of the diagonal-block inverses of the two-level factorisation (batches of matrices)
    double *w = nullptr, *alpha = nullptr, *hyper = nullptr, *jitter = nullptr, *nll = nullptr;
    int* info = nullptr;
    int* flags = nullptr;   // hand-off words of the chain kernel, [batch][chain_flag_count(Np/64)]
    int inv_panels = 0;     // host-side: 0 = Inv holds L^-1; W > 0: only the inverses of the diagonal blocks of W block columns
                            // (a value-only factorisation: factor_twolevel(want_inverse = false); twolevel_inverse_all completes it)
    long mat() const { return (long)Np * Np; }
    // scratch of the triangular inverse per matrix: [0, hw^2) level scratch, then one slot per high-level node
    long hw() const { return Np / 2 + 64; }
    static long wl_stride() { return 512L * 512L; }     // super-panels of up to 16 block columns: (32 W)^2 doubles per matrix
    long wstride() const {
        long slots = 0;                 // sum of h2 * s over the nodes above the segment level (trtri_segment)
        for (long s = SEGR; s < Np; s *= 2)
            for (long base = 0; base + s < Np; base += 2 * s) slots += std::min(s, Np - base - s) * s;
        return hw() * hw() + slots;
    }
};

static int ws_alloc(Workspace& ws, int batch, int Np, int d) {
    ws.batch = batch;
    ws.Np = Np;
    ws.d = d;
    const size_t mb = (size_t)batch * Np * Np * sizeof(double);
    HIPCHK(block_alloc(&ws.K, mb));
    HIPCHK(block_alloc(&ws.L, mb));
    HIPCHK(block_alloc(&ws.Inv, mb));
    HIPCHK(block_alloc(&ws.W, (size_t)batch * ws.wstride() * sizeof(double)));
    HIPCHK(hipMalloc(&ws.Wl, (size_t)batch * Workspace::wl_stride() * sizeof(double)));
    HIPCHK(hipMalloc(&ws.w, (size_t)batch * Np * sizeof(double)));
    HIPCHK(hipMalloc(&ws.alpha, (size_t)batch * Np * sizeof(double)));
    HIPCHK(hipMalloc(&ws.hyper, (size_t)batch * (d + 2) * sizeof(double)));
    HIPCHK(hipMalloc(&ws.jitter, (size_t)batch * sizeof(double)));
    HIPCHK(hipMalloc(&ws.nll, (size_t)batch * sizeof(double)));
    HIPCHK(hipMalloc(&ws.info, (size_t)batch * sizeof(int)));
    HIPCHK(hipMalloc(&ws.flags, (size_t)batch * chain_flag_count(Np / 64) * sizeof(int)));
    HIPCHK(hipMemset(ws.K, 0, mb));
    HIPCHK(hipMemset(ws.L, 0, mb));
    HIPCHK(hipMemset(ws.Inv, 0, mb));
    HIPCHK(hipMemset(ws.alpha, 0, (size_t)batch * Np * sizeof(double)));
    HIPCHK(hipMemset(ws.w, 0, (size_t)batch * Np * sizeof(double)));
    // only factor_with_jitter writes these; gpmpc_set_factors -> gpmpc_append reads jitter without a fit in between
    HIPCHK(hipMemset(ws.jitter, 0, (size_t)batch * sizeof(double)));
    HIPCHK(hipMemset(ws.nll, 0, (size_t)batch * sizeof(double)));
    HIPCHK(hipMemset(ws.info, 0, (size_t)batch * sizeof(int)));
    return GPMPC_OK;
}

static void ws_free(Workspace& ws) {
    block_free(ws.K); block_free(ws.L); block_free(ws.Inv); block_free(ws.InvK); block_free(ws.W);
    hipFree(ws.Wl);
    hipFree(ws.w); hipFree(ws.alpha); hipFree(ws.hyper); hipFree(ws.jitter); hipFree(ws.nll); hipFree(ws.info); hipFree(ws.flags);
    ws = Workspace();
}

static int ws_need_invK(Workspace& ws) {
    if (!ws.InvK) HIPCHK(block_alloc(&ws.InvK, (size_t)ws.batch * ws.mat() * sizeof(double)));
    return GPMPC_OK;
}

struct Prof {
    bool on = false;
    unsigned mask = ~0u;            // phases that are bracketed while `on` (bit = phase index)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[GPMPC_PH_COUNT];
    std::vector<hipEvent_t> pool;
    double total[GPMPC_PH_COUNT] = {0};
    long count[GPMPC_PH_COUNT] = {0};
};

// HIP-event bracket of one phase on a stream (gpmpc_profile_*); inert unless profiling is on
struct ProfScope {
    Prof* pr;
    hipStream_t st;
    int phase;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(Prof* pr_, hipStream_t st_, int ph) : pr(pr_), st(st_), phase(ph) {
        if (!pr || !pr->on || !((pr->mask >> ph) & 1u)) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!pr->pool.empty()) { e = pr->pool.back(); pr->pool.pop_back(); }
            else hipEventCreate(&e);
            return e;
        };
        e0 = get();
        e1 = get();
        hipEventRecord(e0, st);
    }
    void end_on(hipStream_t s) { st = s; }     // the bracket closes on another queue (a phase that spans two)
    ~ProfScope() {
        if (!e0) return;
        hipEventRecord(e1, st);
        pr->ev[phase].push_back({e0, e1});
    }
};

// What a chained factorisation leaves for its caller and for the first prediction behind it.
//  * Early status: `info` and the hand-off words are final when the chain kernel ends, ~0.3 ms before the last rows of
//    L^-1.  factor_chain copies them to the host on the workers' queue (idle by then) and records `ev_info` there; the
//    fit returns on that event with the tail of the inverse still in flight on the main queue.
//  * alpha off the main queue: gpmpc_fit then forms alpha on the workers' queue behind the tail (ev_alpha); the main
//    queue waits for it in front of the next consumer (alpha_ready), so a variance product that does not read alpha
//    follows the tail directly.
//  * The first prediction behind such a fit (`armed`) forms its cross-covariances on the low-priority queue while the
//    tail runs, and its mean (which needs alpha) next to the variance product (predict_chunk).
struct TailState {
    bool armed = false;          // set by gpmpc_fit, consumed (or dropped) by the next call that touches the predict scratch
    bool alpha_pending = false;  // alpha of the model workspace is being formed on the workers' queue: wait for ev_alpha
    hipEvent_t ev_chain = nullptr, ev_tail = nullptr, ev_alpha = nullptr, ev_ks = nullptr, ev_mean = nullptr;
    hipEvent_t ev_w = nullptr;   // w = L^-1 y of the pending alpha is there (the fused mean of the variance product needs no more)
    // early status (set up by factor_with_jitter per attempt)
    int* pin_info = nullptr;
    int* cerr = nullptr;
    size_t nflag = 0;
    int nb = 0;
    hipEvent_t ev_info = nullptr;
    bool want_early = false, early_done = false;
    bool fused_early = false;    // gpmpc_fit_predict_mean_var: a prediction is being enqueued in front of the host's wait for the status words
    static hipEvent_t get(hipEvent_t& e) {
        if (!e) hipEventCreateWithFlags(&e, hipEventDisableTiming);
        return e;
    }
};

struct Ctx {
    hipStream_t stream;
    int crow_mode;
    hipStream_t side = nullptr;     // second queue for the bulk work of the chained factorisation
    hipEvent_t fork = nullptr, join = nullptr;
    hipStream_t aux = nullptr;      // third queue: pipelined pieces of the triangular inverse
    hipEvent_t* seg = nullptr;      // pool of n_seg events (segment hand-offs side -> aux, aux -> main)
    int n_seg = 0;
    int workers = 0;                // > 0: tile-owner worker kernel with this many CUs to share (chain mode 3)
    Prof* prof = nullptr;           // the handle's profile (phase brackets inside the factorisation)
    hipStream_t bulk = nullptr;     // fourth queue (low priority): look-ahead part of the two-level trailing updates
    TailState* tail = nullptr;      // early status + row-panel events of the chained factorisation (may be null)
    bool value_only = false;        // the caller needs L and the diagonal blocks' inverses only (a line-search trial): where the
                                    // execution can, it leaves L^-1 unformed and says so in ws.inv_panels
    bool no_workers = false;        // never the tile-owner workers: the choice of execution must not depend on the batch size
                                    // (lock-step restart search: a point's value may not depend on what else is in its batch)
};
