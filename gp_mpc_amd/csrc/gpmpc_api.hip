// C ABI of libgpmpc_hip.so (include/gpmpc.h): host-side orchestration of the HIP kernels.
// One translation unit: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC gpmpc_api.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gpmpc.h"
#include "chol_chain.hpp"
#include "chol_worker.hpp"
#include "em_kernels.hpp"
#include "gemm_f64_dma.hpp"
#include "gp_kernels.hpp"
#include "leaf64.hpp"
#include "train_native.hpp"

using namespace gpmpc;

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
    double *w = nullptr, *alpha = nullptr, *hyper = nullptr, *jitter = nullptr, *nll = nullptr;
    int* info = nullptr;
    int* flags = nullptr;   // hand-off words of the chain kernel, [batch][chain_flag_count(Np/64)]
    long mat() const { return (long)Np * Np; }
    // scratch of the triangular inverse per matrix: [0, hw^2) level scratch, then one slot per high-level node
    long hw() const { return Np / 2 + 64; }
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
    hipFree(ws.w); hipFree(ws.alpha); hipFree(ws.hyper); hipFree(ws.jitter); hipFree(ws.nll); hipFree(ws.info); hipFree(ws.flags);
    ws = Workspace();
}

static int ws_need_invK(Workspace& ws) {
    if (!ws.InvK) HIPCHK(block_alloc(&ws.InvK, (size_t)ws.batch * ws.mat() * sizeof(double)));
    return GPMPC_OK;
}

struct Prof {
    bool on = false;
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
        if (!pr || !pr->on) return;
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
    ~ProfScope() {
        if (!e0) return;
        hipEventRecord(e1, st);
        pr->ev[phase].push_back({e0, e1});
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
};

static GemmP gemm_base(const Ctx& cx) {
    GemmP p;
    std::memset(&p, 0, sizeof(p));
    p.alpha = 1.0;
    p.crow_mode = cx.crow_mode;
    return p;
}

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
static void trtri_range(const Ctx& cx, Workspace& ws, hipStream_t stream, long base0, int n) {
    const long ld = ws.Np, sM = ws.mat(), sW = ws.wstride();
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
            t.C = ws.W; t.ldc = s;
            t.M = h2; t.N = s; t.K = s;
            t.zdiv = nodes; t.sA = snode; t.sB = snode; t.sC = (long)s * s; t.sA2 = sM; t.sB2 = sM; t.sC2 = sW;
            launch_gemm(t, nodes * ws.batch, stream);
            GemmP u = gemm_base(cx);                    // inv21 = -inv22 W
            u.A = ws.Inv + o22; u.lda = ld; u.a_mc = 0; u.kflags = KA_LE_M;
            u.B = ws.W; u.ldb = s; u.b_nc = 1;
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
static bool factor_twolevel(const Ctx& cx, Workspace& ws, int spin_limit, int W) {
    const int Np = ws.Np, nb = Np / 64, nf = chain_flag_count(nb);
    const long ld = Np, sM = ws.mat(), sW = ws.wstride();
    int* leafdone = ws.flags + 1;
    int* pan1 = ws.flags + 1 + nb;
    int* tdone = ws.flags + 1 + 2 * nb;
    // The inverse follows panel by panel on the third queue -- right-looking blocked inversion of the row panels
    // P_i = super-panel i: with S = sum over finished panels m of L[., P_m] X[P_m, .] accumulated IN the not yet final
    // rows of Inv,
    //     I_i = (L[P_i, P_i])^-1 (level-batched),    X[P_i, < r_i] = -I_i S[P_i, < r_i],
    //     S[> P_i, < r_{i+1}] += L[> P_i, P_i] X[P_i, < r_{i+1}]                  (K = 64 W products)
    // so every step only needs rows P_i of L -- final as soon as super-panel i is factored -- and after the last
    // super-panel just its own inverse and one 64 W-row product remain (the tree-shaped inverse left the two products of
    // its root, a third of the fit, for the end).  Needs a 64 W x Np scratch panel in ws.W and the event pool.
    const bool panel_inv = cx.aux && cx.seg && cx.n_seg >= 3 && (long)64 * W * Np <= sW;
    auto inverse_panel = [&](hipStream_t st, int k0, int k1) {
        const int ri = 64 * k0, a = 64 * (k1 - k0), rn = 64 * k1, Mb = Np - rn;
        trtri_range(cx, ws, st, ri, a);                                        // I_i
        if (ri > 0) {
            GemmP u = gemm_base(cx);                                           // T = -I_i S_i, then back into Inv[P_i, < r_i]
            u.A = ws.Inv + (long)ri * ld + ri; u.lda = ld; u.sA = sM; u.a_mc = 0; u.kflags = KA_LE_M;
            u.B = ws.Inv + (long)ri * ld; u.ldb = ld; u.sB = sM; u.b_nc = 1;
            u.C = ws.W; u.ldc = ri; u.sC = sW;
            u.M = a; u.N = ri; u.K = a; u.alpha = -1.0;
            launch_gemm(u, ws.batch, st);
            for (int b = 0; b < ws.batch; ++b)
                hipMemcpy2DAsync(ws.Inv + b * sM + (long)ri * ld, ld * sizeof(double), ws.W + b * sW, (size_t)ri * sizeof(double),
                                 (size_t)ri * sizeof(double), a, hipMemcpyDeviceToDevice, st);
        }
        if (Mb > 0) {
            GemmP t = gemm_base(cx);                                           // new columns of S: L[> P_i, P_i] I_i
            t.A = ws.L + (long)rn * ld + ri; t.lda = ld; t.sA = sM; t.a_mc = 0;
            t.B = ws.Inv + (long)ri * ld + ri; t.ldb = ld; t.sB = sM; t.b_nc = 1; t.kflags = KB_GE_N;
            t.C = ws.Inv + (long)rn * ld + ri; t.ldc = ld; t.sC = sM;
            t.M = Mb; t.N = a; t.K = a;
            launch_gemm(t, ws.batch, st);
            if (ri > 0) {
                GemmP v = gemm_base(cx);                                       // S[> P_i, < r_i] += L[> P_i, P_i] X[P_i, < r_i]
                v.A = ws.L + (long)rn * ld + ri; v.lda = ld; v.sA = sM; v.a_mc = 0;
                v.B = ws.Inv + (long)ri * ld; v.ldb = ld; v.sB = sM; v.b_nc = 1;
                v.C = ws.Inv + (long)rn * ld; v.ldc = ld; v.sC = sM;
                v.M = Mb; v.N = ri; v.K = a; v.beta = 1.0;
                launch_gemm(v, ws.batch, st);
            }
        }
    };
    int ev = 0;                                                // event pool cursor
    int inv_done = 0;                                          // block columns whose inverse panel has been enqueued
    // Look-ahead: the K = 64 W update of super-panel s is split in A(s) = the NEXT super-panel's columns (second queue,
    // what the chain needs next) and B(s) = everything right of them (fourth queue, low priority), so that B(s) overlaps
    // the latency-bound factorisation of super-panel s+1.  Order on shared tiles: A(s) after B(s-1) (event), B(s) after
    // the panels of s (event) and after B(s-1) (queue order).
    static const bool lookahead_on = !(getenv("GPMPC_LOOKAHEAD") && atoi(getenv("GPMPC_LOOKAHEAD")) == 0);
    const bool lookahead = lookahead_on && cx.bulk && cx.seg && cx.n_seg >= 4 * ((nb + W - 1) / W) + 2;
    hipEvent_t evB_prev = nullptr;
    if (lookahead) {
        hipEventRecord(cx.join, cx.stream);                    // the fourth queue starts behind everything enqueued so far
        hipStreamWaitEvent(cx.bulk, cx.join, 0);
    }
    for (int k0 = 0; k0 < nb; k0 += W) {
        const int k1 = std::min(nb, k0 + W), k2 = std::min(nb, k1 + W);
        // the chain of this super-panel starts when the update of its columns (second queue) is complete
        hipEventRecord(cx.join, cx.side);
        hipStreamWaitEvent(cx.stream, cx.join, 0);
        hipLaunchKernelGGL(chol_chain_kernel, dim3(1, 1, ws.batch), dim3(256), CHAIN_LDS_BYTES, cx.stream, (const double*)ws.K,
                           ws.L, ws.Inv, ld, sM, nb, ws.flags, (long)nf, ws.info, cx.crow_mode, spin_limit, g_chain_trace, 0, k0,
                           k1);
        hipLaunchKernelGGL(flag_gate_kernel, dim3(ws.batch), dim3(64), 0, cx.side, ws.flags, (long)nf, 1 + k0, 1, -1, 0,
                           spin_limit);                       // bulk workgroups only once this chain launch is resident
        for (int k = k0; k < k1; ++k) {
            const int off = 64 * k;
            const long o11 = (long)off * ld + off;
            const bool last = k + 1 == k1;                     // the chain stops after this leaf: row k+1 is the panel product's
            const int r0 = off + (last ? 64 : 128), M2 = Np - r0;
            if (M2 > 0) {
                GemmP p = gemm_base(cx);                       // panel: L(i,k) = A(i,k) inv_kk^T
                p.A = ws.K + (long)r0 * ld + off; p.lda = ld; p.sA = sM; p.a_mc = 0;
                p.B = ws.Inv + o11; p.ldb = ld; p.sB = sM; p.b_nc = 0; p.kflags = KB_LE_N;
                p.C = ws.L + (long)r0 * ld + off; p.ldc = ld; p.sC = sM;
                p.M = M2; p.N = 64; p.K = 64;
                p.wait_flag = leafdone + k; p.err = ws.flags; p.spin_limit = spin_limit; p.sFlags = nf;
                launch_gemm(p, ws.batch, cx.side);
            }
            const int M1 = Np - off - 64, N1 = 64 * (k1 - k - 1);   // trailing update inside the super-panel's columns
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
        // rows < 64 k1 of L are final: the inverse of this row panel goes to the third queue, next to the big update
        if (panel_inv && k1 < nb && ev + 2 < cx.n_seg) {
            hipEventRecord(cx.seg[ev], cx.side);
            hipStreamWaitEvent(cx.aux, cx.seg[ev], 0);
            ++ev;
            hipEventRecord(cx.seg[ev], cx.stream);            // (the leaf's own stores: the chain launch has to be complete)
            hipStreamWaitEvent(cx.aux, cx.seg[ev], 0);
            ++ev;
            if (inv_done < k0) inverse_panel(cx.aux, inv_done, k0);   // (panels skipped for want of events: as one)
            inverse_panel(cx.aux, k0, k1);
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
    if (!panel_inv) { trtri_levels(cx, ws); return true; }
    inverse_panel(cx.stream, inv_done, nb);                    // what is left: the last panel (or everything not handed over)
    return true;
}

// Chained factorisation: the sequential part of every panel step runs in ONE persistent workgroup
// (chol_chain_kernel, main queue) that keeps a CU to itself, the bulk -- panel rows >= k+2 and the
// trailing update -- in ordinary GEMM launches on the side queue; flags in ws.flags couple the two.
// Returns false if the path is unavailable (no side queue).  A time-out inside the kernels is reported
// through ws.flags[0] and handled by the caller (fallback to factor_blocked).

static bool factor_chain(const Ctx& cx, Workspace& ws, int spin_limit) {
    if (!cx.side || ws.Np < 128) return false;
    const int Np = ws.Np, nb = Np / 64, nf = chain_flag_count(nb);
    const long ld = Np, sM = ws.mat();
    hipMemsetAsync(ws.flags, 0, (size_t)ws.batch * nf * sizeof(int), cx.stream);
    hipEventRecord(cx.fork, cx.stream);
    hipStreamWaitEvent(cx.side, cx.fork, 0);
    // bulk work as tile-owner workers: 7 of 8 CUs run one, the trailing matrix lives in their registers
    // A worker fills a CU (512 threads x ~250 VGPRs) and the chain needs an empty CU too.  Measured on MI355X
    // (start-time stamps of the workers): workgroups are dealt to the shader engines (8 CUs each) in a fixed
    // rotation and a workgroup that does not fit on "its" engine waits there even when CUs are free elsewhere
    // -- with 8 workers on the engine that also got the chain, the 8th started 234 ms late, after the others'
    // polls had timed out.  So: 7 workers per engine, nothing else in flight but the chain (one matrix only).
    const int ntiles = (nb - 1) * nb / 2 - 1;            // tiles kept in registers (chol_worker.hpp)
    int NW = ws.batch == 1 ? cx.workers - cx.workers / 8 : 0;
    if (NW > ntiles) NW = ntiles;
    const bool use_workers = NW >= 1 && nb >= 3 && (ntiles + NW - 1) / NW <= WORKER_MAXT;
    // what the workers do not take: two-level panels (GPMPC_TWOLEVEL=<block columns per super-panel>, 0/1 = off)
#ifdef GPMPC_EMULATED
    static const int twolevel_W = getenv("GPMPC_TWOLEVEL") ? atoi(getenv("GPMPC_TWOLEVEL")) : 2;
#else
    static const int twolevel_W = getenv("GPMPC_TWOLEVEL") ? atoi(getenv("GPMPC_TWOLEVEL")) : 8;
#endif
    if (!use_workers && twolevel_W > 1 && nb >= 2 * twolevel_W && cx.aux && cx.seg) {
        static const bool verbose2 = getenv("GPMPC_VERBOSE") != nullptr;
        if (verbose2)
            fprintf(stderr, "gpmpc: factor Np=%d batch=%d: two-level panels of %d block columns\n", Np, ws.batch, twolevel_W);
        return factor_twolevel(cx, ws, spin_limit, twolevel_W);
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
            if (nt > 0 && nw > nt) nw = nt;
            if (a < SEGR || nbr < 3 || nt < 1 || (nt + nw - 1) / nw > WORKER_MAXT) break;
            r[L] = start; nws[L] = nw; ++L;
            int nxt = 64;                                   // next cut: the left child of what remains
            while (2 * nxt < Np - start) nxt *= 2;
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
    {   // the chain kernel ends with the last leaf, i.e. when L is complete: its duration is the Cholesky's
        ProfScope t(cx.prof, cx.stream, GPMPC_PH_CHAIN);
        hipLaunchKernelGGL(chol_chain_kernel, dim3(1, 1, ws.batch), dim3(256), CHAIN_LDS_BYTES, cx.stream, (const double*)ws.K,
                           ws.L, ws.Inv, ld, sM, nb, ws.flags, (long)nf, ws.info, cx.crow_mode, spin_limit, g_chain_trace,
                           use_workers ? 1 : 0);
    }
    static const bool verbose = getenv("GPMPC_VERBOSE") != nullptr;
    if (verbose)
        fprintf(stderr, "gpmpc: factor Np=%d batch=%d: chain kernel + %s (%d launch%s), inverse %s\n", Np, ws.batch,
                use_workers ? "tile-owner workers" : "GEMM launches", use_workers ? L : 0, L == 1 ? "" : "es",
                split ? "by row panels behind the worker launches" : pipelined ? "pipelined" : "at the end");
    if (use_workers) {
        for (int i = 0; i < L; ++i) {
            int* ready = i ? ws.flags + chain_ready_index(nb) + 2 * (i - 1) : nullptr;   // arrival counter + flag of launch i
            hipLaunchKernelGGL(chol_worker_kernel, dim3(nws[i], 1, ws.batch), dim3(WORKER_THREADS), WORKER_LDS_BYTES, cx.side,
                               ws.K, ws.L, (const double*)ws.Inv, ld, sM, nb, ws.flags, (long)nf, cx.crow_mode, spin_limit,
                               r[i] / 64, i + 1 < L ? (r[i + 1] - r[i]) / 64 : nb, ready, g_chain_trace);
            if (i + 1 == L) break;
            // launch i finished: rows P_i of L are final.  Behind launch i + 1, once it is resident:
            hipEventRecord(cx.seg[i], cx.side);
            hipStreamWaitEvent(cx.aux, cx.seg[i], 0);
            hipLaunchKernelGGL(flag_gate_kernel, dim3(ws.batch), dim3(64), 0, cx.aux, ws.flags, (long)nf,
                               chain_ready_index(nb) + 2 * i + 1, 1, -1, 0, spin_limit);
            const int ri = r[i], a = r[i + 1] - r[i];
            trtri_range(cx, ws, cx.aux, ri, a);                                    // I_i
            if (i + 2 == L) hipEventRecord(cx.seg[cx.n_seg - 2], cx.aux);         // the side queue's last use of the level scratch
            const double* Ii = ws.Inv + (long)ri * ld + ri;
            const double* Si = i ? ws.W + wofs[i] : nullptr;                       // a x ri
            for (int jj = i + 1; jj < L; ++jj) {
                const int rj = r[jj], hj = r[jj + 1] - r[jj];
                double* Sj = ws.W + wofs[jj];
                product(cx.aux, ws.L + (long)rj * ld + ri, ld, KB_GE_N, Ii, ld, Sj + ri, rj, hj, a, a, 1.0, 0.0);        // W_j
                if (i) product(cx.aux, Sj + ri, rj, 0, Si, ri, Sj, rj, hj, ri, a, -1.0, 1.0);                            // S_j -= W_j S_i
            }
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
    if (split) {                                            // the last panel: its own inverse, then -I S
        // The inverse of the last panel needs nothing from the side queue but the level scratch, which that queue left
        // long ago (event recorded behind its last trtri_range); only the product waits for its S.  (Waiting for the
        // whole side queue first put its last product, which ends ~50 us after the chain, in front of these eight
        // latency-bound launches.)
        hipStreamWaitEvent(cx.stream, cx.seg[cx.n_seg - 2], 0);
        const int rl = r[L - 1], h = Np - rl;
        trtri_range(cx, ws, cx.stream, rl, h);
        hipEventRecord(cx.seg[cx.n_seg - 1], cx.aux);
        hipStreamWaitEvent(cx.stream, cx.seg[cx.n_seg - 1], 0);
        product(cx.stream, ws.Inv + (long)rl * ld + rl, ld, KA_LE_M, ws.W + wofs[L - 1], rl, ws.Inv + (long)rl * ld, ld,
                h, rl, h, -1.0, 0.0);
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
static void solve_alpha(const Ctx& cx, Workspace& ws, const double* y, long sy) {
    const int Np = ws.Np;
    hipLaunchKernelGGL(gemv_rows_kernel, dim3(Np / 4, ws.batch), dim3(256), 0, cx.stream, ws.Inv, y, ws.w, Np, ws.mat(), sy,
                       (long)Np, 1);
    const int chunks = (Np + GEMVT_ROWS - 1) / GEMVT_ROWS;         // partial sums go through the (now idle) inverse scratch
    hipLaunchKernelGGL(gemv_lowerT_part_kernel, dim3((Np + 127) / 128, chunks, ws.batch), dim3(256), 0, cx.stream, ws.Inv, ws.w, ws.W,
                       Np, ws.mat(), (long)Np, ws.wstride());
    hipLaunchKernelGGL(gemv_lowerT_finish_kernel, dim3((Np + 255) / 256, ws.batch), dim3(256), 0, cx.stream, ws.W, ws.alpha, Np,
                       chunks, ws.wstride(), (long)Np);
}

// K^-1 = L^-T L^-1 (lower triangle by MFMA, then mirrored)
static int compute_invK(const Ctx& cx, Workspace& ws) {
    CHK(ws_need_invK(ws));
    const long ld = ws.Np, sM = ws.mat();
    GemmP p = gemm_base(cx);
    p.A = ws.Inv; p.lda = ld; p.sA = sM; p.a_mc = 1;
    p.B = ws.Inv; p.ldb = ld; p.sB = sM; p.b_nc = 1;
    p.kflags = KA_GE_M | KB_GE_N;
    p.C = ws.InvK; p.ldc = ld; p.sC = sM;
    p.M = ws.Np; p.N = ws.Np; p.K = ws.Np; p.lower = 1;
    launch_gemm(p, ws.batch, cx.stream);
    hipLaunchKernelGGL(symmetrize_kernel, dim3(ws.Np / 64, ws.Np / 64, ws.batch), dim3(256), 0, cx.stream, ws.InvK,
                       ws.Np);
    return GPMPC_OK;
}

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
    Ctx cx() {
        return Ctx{stream, crow_mode, side_stream, ev_fork, ev_join, chain_mode >= 2 ? aux_stream : nullptr,
                   seg_events.data(), (int)seg_events.size(), chain_mode >= 3 ? g_cu_count[device] : 0, &prof,
                   chain_mode >= 2 ? bulk_stream : nullptr};
    }
};

struct PhaseTimer : ProfScope {
    PhaseTimer(gpmpc_gp* h, int ph) : ProfScope(&h->prof, h->stream, ph) {}
};

static int prof_collect(gpmpc_gp* h) {
    HIPCHK(hipStreamSynchronize(h->stream));
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
// super-panel of the two-level execution (>= 2 block columns each; two more each with the look-ahead)
static size_t seg_event_count(int Np) { return (size_t)std::max(3, std::max(Np / SEGR + 2, 2 * (Np / 64) + 8)); }

static int create_impl(gpmpc_gp* h, const double* X, const double* Y) {
    const int device = h->device, N = h->N, d = h->d, Ny = h->Ny;
    h->crow_mode = g_crow_mode[device];
    HIPCHK(hipStreamCreate(&h->own_stream));
    h->stream = h->own_stream;
    HIPCHK(hipStreamCreate(&h->side_stream));
    HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    HIPCHK(hipStreamCreate(&h->aux_stream));
    {
        int lo = 0, hi = 0;                                    // (numerically larger = lower priority)
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&h->bulk_stream, hipStreamDefault, lo));
    }
    // the persistent kernels ask for more than the default 64 KB of dynamic LDS (per device: set for every handle)
    HIPCHK(hipFuncSetAttribute((const void*)chol_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CHAIN_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)chol_worker_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WORKER_LDS_BYTES));
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
    ws_free(h->ws);
    ws_free(h->tws);
    hipFree(h->XT); hipFree(h->Y); hipFree(h->gradPartial); hipFree(h->gradOut);
    hipFree(h->mpar); hipFree(h->Yc); hipFree(h->tmpar); hipFree(h->tYc);
    hipFree(h->Z); hipFree(h->Sigma); hipFree(h->KsT); hipFree(h->part); hipFree(h->meanT);
    hipFree(h->mean); hipFree(h->var); hipFree(h->J); hipFree(h->cov); hipFree(h->em); hipFree(h->ems);
    hipFree(h->beta); hipFree(h->UT); hipFree(h->VT); hipFree(h->sensH); hipFree(h->sensV); hipFree(h->ccpart);
    for (int ph = 0; ph < GPMPC_PH_COUNT; ++ph)
        for (auto& pr : h->prof.ev[ph]) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto e : h->prof.pool) hipEventDestroy(e);
    if (h->ev_info) hipEventDestroy(h->ev_info);
    if (h->pin) hipHostFree(h->pin);
    if (h->io_pin) hipHostFree(h->io_pin);
    hipFree(h->io_dev);
    drop_roll_graphs(h);
    if (h->roll_pin) hipHostFree(h->roll_pin);
    hipFree(h->roll_dev);
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
    h->hyper.assign((size_t)h->Ny * h->nh(), 0.0);     // rows change width: the model has to be fitted / loaded again
    h->fitted = false;
    h->have_invK = false;
    h->have_beta = false;
    return GPMPC_OK;
}

int gpmpc_set_hyper_prior(gpmpc_gp* h, const double* prior6) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    h->have_prior = prior6 != nullptr;
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
    else if (std::strcmp(name, "workspace_blocks_reused") == 0 || std::strcmp(name, "workspace_blocks_fresh") == 0) {
        std::lock_guard<std::mutex> lk(g_block_mutex);
        *value = std::strcmp(name, "workspace_blocks_reused") == 0 ? g_block_reuses : g_block_fresh;
    }
    else return fail(GPMPC_EINVAL, "unknown counter '%s'", name);
    return GPMPC_OK;
}

int gpmpc_profile_enable(gpmpc_gp* h, int enable) {
    if (!h) return fail(GPMPC_EINVAL, "NULL handle");
    h->prof.on = enable != 0;
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

// ------------------------------------------------------------------------------------------------
// fit
// ------------------------------------------------------------------------------------------------
// gram + factor on a workspace whose hyper/jitter buffers are already on the device.
static void gram_and_factor(gpmpc_gp* h, Workspace& ws) {
    const Ctx cx = h->cx();
    {
        PhaseTimer t(h, GPMPC_PH_GRAM);
        launch_gram(cx.stream, dim3(ws.Np / 64, ws.Np / 64, ws.batch), h->d, h->XT, ws.hyper, ws.jitter, ws.K, h->N, ws.Np);
    }
    {
        PhaseTimer t(h, GPMPC_PH_FACTOR);
        hipMemsetAsync(ws.info, 0, ws.batch * sizeof(int), cx.stream);
        if (!(h->chain_mode && factor_chain(cx, ws, h->spin_limit))) factor_blocked(cx, ws, true);
    }
}

// Runs gram+Cholesky with the reference's one-shot jitter rule (optimize.py:345-350).
// info_out[b]: 0 ok, 1 jitter applied, <0: -(first bad pivot) after jitter.
// `post` enqueues the work that consumes the factors (alpha, K^-1, the NLL terms).  It goes into the stream right
// after the copies of the status words and BEFORE the host waits for them -- the host waits on an event recorded
// between the two -- so the host's round trip (wake up, inspect, return to the caller, next launches: ~50 us)
// overlaps with that work instead of leaving the device idle.  If the attempt turns out to have failed (jitter rule,
// hand-off time-out) the next attempt overwrites what `post` produced.
static int factor_with_jitter(gpmpc_gp* h, Workspace& ws, const double* hyper_host, int* info_out,
                              const std::function<void()>& post = std::function<void()>()) {
    const int nb = ws.batch;
    std::vector<double> jit(nb, 0.0);
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
    if (h->chain_mode) turn.lock();               // held until the status words are back, i.e. the factorisation is done
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIPCHK(hipMemcpyAsync(ws.jitter, jit.data(), nb * sizeof(double), hipMemcpyHostToDevice, h->stream));
        gram_and_factor(h, ws);
        HIPCHK(hipGetLastError());
        const bool check_chain = h->chain_mode && h->side_stream && ws.Np >= 128;
        HIPCHK(hipMemcpyAsync(pin_info, ws.info, nb * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        if (check_chain) HIPCHK(hipMemcpyAsync(cerr, ws.flags, nflag * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipEventRecord(h->ev_info, h->stream));
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
                HIPCHK(hipStreamSynchronize(h->stream));
                HIPCHK(hipStreamSynchronize(h->side_stream));
                if (h->aux_stream) HIPCHK(hipStreamSynchronize(h->aux_stream));
                if (h->bulk_stream) HIPCHK(hipStreamSynchronize(h->bulk_stream));
                gram_and_factor(h, ws);
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
                if (attempt == 0) { jit[b] = 1e-8; res[b] = 1; }
                else res[b] = -info[b];
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

extern "C" int gpmpc_fit(gpmpc_gp* h, const double* hyper, int want_invK, int* info) {
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
    int post_rc = GPMPC_OK;
    std::vector<double> kpart;           // [Ny][d+2]; y - m(X) goes to h->Yc (optimize.py:285,494)
    CHK(upload_mean_and_residual(h, hyper, h->Ny, kpart, &h->mpar, h->Y, &h->Yc));
    CHK(factor_with_jitter(h, h->ws, kpart.data(), info, [&]() {
        {
            PhaseTimer t(h, GPMPC_PH_SOLVE);
            solve_alpha(h->cx(), h->ws, h->y_model(), h->Np);
        }
        if (want_invK) {
            PhaseTimer t(h, GPMPC_PH_INVK);
            post_rc = compute_invK(h->cx(), h->ws);
        }
    }));
    CHK(post_rc);
    if (want_invK) h->have_invK = true;
    HIPCHK(hipGetLastError());
    h->hyper.assign(hyper, hyper + (size_t)h->Ny * nh);
    h->fitted = true;
    return GPMPC_OK;
}

// ---- data update: a15 (GP.update_data_all gp_class.py:474-550 = append + full recomputation with the
// existing hyper-parameters) as a rank-n extension of the factors.
// With R0 = 64 floor(N/64) the rows < R0 of L and L^-1 do not change.  For the strip of m = Np' - R0 rows
// below (the last partial block of old points, the new points, padding):
//     K' rows >= R0 from the K build;   L21 = K21 inv11^T;   S = K22 - L21 L21^T;   L22 = chol(S) (blocked);
//     inv22 = L22^-1;   inv21 = -inv22 (L21 inv11)
// i.e. four GEMMs with K = R0 plus a factorisation of m rows -- O(N^2 m) instead of O(N^3).
static void free_predict_scratch(gpmpc_gp* h) {
    hipFree(h->Z); hipFree(h->Sigma); hipFree(h->KsT); hipFree(h->part); hipFree(h->meanT);
    hipFree(h->mean); hipFree(h->var); hipFree(h->J); hipFree(h->cov); hipFree(h->UT); hipFree(h->VT);
    hipFree(h->sensH); hipFree(h->sensV); hipFree(h->em); hipFree(h->ems); hipFree(h->beta); hipFree(h->gradPartial); hipFree(h->gradOut);
    hipFree(h->ccpart); hipFree(h->Yc); hipFree(h->tYc);
    h->Yc = h->tYc = nullptr;
    h->Z = h->Sigma = h->KsT = h->part = h->meanT = h->mean = h->var = h->J = h->cov = h->UT = h->VT = nullptr;
    h->sensH = h->sensV = h->em = h->ems = h->beta = h->gradPartial = h->gradOut = h->ccpart = nullptr;
    h->Bcap = 0;
    h->emBytes = h->emsBytes = 0;
    h->have_beta = false;
    ws_free(h->tws);
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

static int ensure_scratch(gpmpc_gp* h, int B) {
    const int need = round_up(B < chunk_size(h) ? B : chunk_size(h), 64);
    if (need <= h->Bcap) return GPMPC_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    hipFree(h->Z); hipFree(h->Sigma); hipFree(h->KsT); hipFree(h->part); hipFree(h->meanT);
    hipFree(h->mean); hipFree(h->var); hipFree(h->J); hipFree(h->cov); hipFree(h->UT); hipFree(h->VT);
    hipFree(h->sensH); hipFree(h->sensV); hipFree(h->ccpart);
    h->UT = h->VT = h->sensH = h->sensV = h->ccpart = nullptr;
    h->Z = h->Sigma = h->KsT = h->part = h->meanT = h->mean = h->var = h->J = h->cov = nullptr;
    h->Bcap = 0;
    const size_t d = h->d, Ny = h->Ny, Np = h->Np, Bc = need;
    HIPCHK(hipMalloc(&h->Z, Bc * d * sizeof(double)));
    HIPCHK(hipMalloc(&h->Sigma, Bc * d * d * sizeof(double)));
    HIPCHK(hipMalloc(&h->KsT, Ny * Bc * Np * sizeof(double)));
    HIPCHK(hipMalloc(&h->part, Ny * (Np / 16) * Bc * sizeof(double)));
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
static int predict_chunk(gpmpc_gp* h, int B, const double* dZ, double* dMean, double* dVar, double* dJ, double* VT = nullptr) {
    const Ctx cx = h->cx();
    const int Bp = round_up(B, 32), Np = h->Np, Ny = h->Ny;
    {
        PhaseTimer t(h, GPMPC_PH_CROSSCOV);
        // few test points (an MPC's shooting nodes): cut the training points in chunks so that the launch fills the chip
        const int nch = (Bp <= CROSSCOV_SMALL_B && Np >= CROSSCOV_CHUNK_MIN_NP) ? CROSSCOV_CHUNKS : 1;
        launch_crosscov(cx.stream, h->d, h->XT, h->ws.hyper, h->ws.alpha, dZ, h->KsT, h->meanT, dJ, h->N, Np, B, Bp, Ny,
                        h->ccpart, nch);
    }
    int tilesM = 0;
    // One point: a dedicated kernel streams L^-1 once at 5.6 TB/s (C3 size).  Measured at N = 8192, Ny = 6
    // (tools/bench_smallb.py), its multi-column versions fall off quickly (B = 2 / 4 / 8: 0.41 / 0.52 / 0.82 ms)
    // while the DMA-staged GEMM below does any B <= 32 in 0.30-0.32 ms: GPMPC_VARSMALL_MAX (default 1) is the switch.
    static const int varsmall_max = getenv("GPMPC_VARSMALL_MAX") ? atoi(getenv("GPMPC_VARSMALL_MAX")) : 1;
    if (dVar && !VT && B <= varsmall_max && B <= 8) {
        PhaseTimer t(h, GPMPC_PH_VARGEMM);   // stream L^-1 once (HBM-bound), no MFMA padding waste
        tilesM = Np / 32;
        const dim3 grid(tilesM, Ny);
        if (B == 1) hipLaunchKernelGGL((var_small_kernel<1>), grid, dim3(256), 0, cx.stream, h->ws.Inv, h->KsT, h->part, Np, Bp);
        else if (B == 2) hipLaunchKernelGGL((var_small_kernel<2>), grid, dim3(256), 0, cx.stream, h->ws.Inv, h->KsT, h->part, Np, Bp);
        else if (B <= 4) hipLaunchKernelGGL((var_small_kernel<4>), grid, dim3(256), 0, cx.stream, h->ws.Inv, h->KsT, h->part, Np, Bp);
        else hipLaunchKernelGGL((var_small_kernel<8>), grid, dim3(256), 0, cx.stream, h->ws.Inv, h->KsT, h->part, Np, Bp);
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
        const int tile = gemm_pick_tile(p, Ny);
        tilesM = (Np + tile - 1) / tile;
        p.sPart = (long)tilesM * Bp;
        launch_gemm(p, Ny, cx.stream, tile);
    }
    {
        PhaseTimer t(h, GPMPC_PH_FINISH);
        hipLaunchKernelGGL(var_finish_kernel, dim3(B), dim3(256), 0, cx.stream, h->part, h->meanT,
                           h->ws.hyper, dMean, dVar, B, Bp, Ny, h->d, tilesM);
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
    CHK(ensure_scratch(h, B));
    const int d = h->d, Ny = h->Ny;
    const bool host = h->ptr_mode == GPMPC_PTR_HOST;
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
            CHK(predict_chunk(h, nb, dZ, oMean, vbuf, jbuf));
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
    if (d > EMK) return fail(GPMPC_EINVAL, "EM: input dimension d=%d exceeds the MFMA cross-term depth %d", d, EMK);
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
            hipLaunchKernelGGL(em_mean_sens_kernel, dim3(Ny, nb), dim3(256), 0, cx.stream, h->XT, dZ, h->beta, prep,
                               o1 + (size_t)b0 * Ny * d, o2 + (size_t)b0 * Ny * d * d, N, Np, d, Ny, b0);
            hipLaunchKernelGGL(em_operands_ordered_kernel, dim3((Np + 255) / 256, PO, nb), dim3(256), 0, cx.stream, h->XT, dZ,
                               h->ws.hyper, prep, h->beta, ops, N, Np, d, Ny, b0);
            hipLaunchKernelGGL(em_pair_sens_kernel<false>, dim3(tiles, PO, nb), dim3(256), 0, cx.stream, ops, h->ws.InvK, h->XT, dZ,
                               part, N, Np, Ny, d, b0, cx.crow_mode);
            hipLaunchKernelGGL(em_pair_sens_kernel<true>, dim3(tiles, PO, nb), dim3(256), 0, cx.stream, ops, h->ws.InvK, h->XT, dZ,
                               part, N, Np, Ny, d, b0, cx.crow_mode);
            hipLaunchKernelGGL(em_sens_reduce_kernel, dim3(PO, nb), dim3(256), 0, cx.stream, part, sums, Ny, tiles);
            hipLaunchKernelGGL(em_sens_finish_kernel, dim3((unsigned)(nb * P)), dim3(DMAX * GJ_LD), 0, cx.stream, sums, prep,
                               h->ws.hyper, dS, oM, o1 + (size_t)b0 * Ny * d, o2 + (size_t)b0 * Ny * d * d,
                               o3 + (size_t)b0 * Ny * Ny * d, o4 + (size_t)b0 * Ny * Ny * d * d, nb, Ny, d, b0);
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

// ------------------------------------------------------------------------------------------------
// a7 NLL (+ analytic gradient) on the separate single-output training workspace
// ------------------------------------------------------------------------------------------------
extern "C" int gpmpc_nll(gpmpc_gp* h, int a, const double* hyper_row, double* nll, double* grad, int* jitter_out) {
    if (!h || !hyper_row || !nll || a < 0 || a >= h->Ny) return fail(GPMPC_EINVAL, "bad arguments");
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
        if (grad) {
            {
                PhaseTimer t(h, GPMPC_PH_INVK);
                GemmP p = gemm_base(cx);  // lower triangle of K^-1 = L^-T L^-1 is all the gradient pass reads
                p.A = ws.Inv; p.lda = Np; p.sA = ws.mat(); p.a_mc = 1;
                p.B = ws.Inv; p.ldb = Np; p.sB = ws.mat(); p.b_nc = 1;
                p.kflags = KA_GE_M | KB_GE_N;
                p.C = ws.InvK; p.ldc = Np; p.sC = ws.mat();
                p.M = Np; p.N = Np; p.K = Np; p.lower = 1;
                launch_gemm(p, 1, cx.stream);
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
    }));
    if (jitter_out) *jitter_out = info;
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(nll, ws.nll, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (grad) HIPCHK(hipMemcpyAsync(grad, h->gradOut, (d + 2 + nmean) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->have_prior) {
        // calc_NLL optimize.py:77-97, literally: `return NLL(...) + log_prior` with log_prior the sum of the Gaussian
        // log-densities prior_gauss(theta, mu, s^2) = -(theta - mu)^2 / (2 s^2) - 1/2 log(2 pi s^2) of every ell_i and of
        // sf^2 and sn^2 (the SQUARED hyper-parameters, :90-91).  (The log-prior is ADDED to the negative log-likelihood
        // there, not subtracted; the reference never enables it, prior = None :157.)
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
        *nll += lp;
    }
    return GPMPC_OK;
}

// ------------------------------------------------------------------------------------------------
// a8 multistart training behind the C ABI (train_gp_numpy optimize.py:359-503 / train_gp :100-294)
// ------------------------------------------------------------------------------------------------
extern "C" int gpmpc_rccl_unique_id(char* id128) {
    if (!id128) return fail(GPMPC_EINVAL, "NULL id buffer");
    RcclApi& R = rccl_api();
    if (!R.ok()) return fail(GPMPC_EHIP, "librccl.so could not be loaded: %s", dlerror() ? dlerror() : "missing symbols");
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
    if (!R.ok()) return fail(GPMPC_EHIP, "librccl.so could not be loaded");
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

extern "C" int gpmpc_train_multistart(gpmpc_gp* h, int nstart, const double* starts, const double* lb, const double* ub,
                                      int max_iter, double tol, int rank, int world, void* rccl_comm, int want_invK,
                                      double* hyper_opt, double* obj, double* theta_all, int* info) {
    if (!h || nstart <= 0 || !starts || !lb || !ub || !hyper_opt) return fail(GPMPC_EINVAL, "NULL argument or nstart <= 0");
    if (world < 1 || rank < 0 || rank >= world) return fail(GPMPC_EINVAL, "bad rank %d / world %d", rank, world);
    HIPCHK(hipSetDevice(h->device));
    const int Ny = h->Ny, nh = h->nh(), d = h->d, row = nh + 1;
    const double inf = std::numeric_limits<double>::infinity();
    if (max_iter <= 0) max_iter = 200;
    if (!(tol > 0.0)) tol = 1e-8;
    std::vector<double> table((size_t)Ny * nstart * row, 0.0);   // [a][r][NLL, theta...]; not-owned / failed: +inf
    int hip_rc = GPMPC_OK;
    for (int a = 0; a < Ny; ++a) {
        BoxProblem P;
        P.n = nh;
        P.lb.assign(lb + (size_t)a * nh, lb + (size_t)(a + 1) * nh);
        P.ub.assign(ub + (size_t)a * nh, ub + (size_t)(a + 1) * nh);
        P.logv.resize(nh);
        for (int k = 0; k < nh; ++k) {
            if (!(P.lb[k] <= P.ub[k])) return fail(GPMPC_EINVAL, "empty box for hyper-parameter %d of output %d", k, a);
            // length scales and sf in log space (their boxes span many decades).  NOT the noise sn: the NLL sees it as sn^2,
            // so d NLL / d log sn = 2 sn^2 (...) vanishes at the reference's start sn = 1e-5 and a log-space search leaves it
            // there -- on the fixture whose optimum has sn on its upper bound it stopped 5.6 above the reference's NLL.
            P.logv[k] = k < d + 1 && P.lb[k] > 0.0 && P.ub[k] < inf;
        }
        P.eval = [&](const double* th, double* f, double* g) -> bool {
            const int rc = gpmpc_nll(h, a, th, f, g, nullptr);
            if (rc == GPMPC_EHIP || rc == GPMPC_ENOMEM) hip_rc = rc;
            return rc == GPMPC_OK;
        };
        for (int r = 0; r < nstart; ++r) {
            double* out = &table[((size_t)a * nstart + r) * row];
            out[0] = inf;
            if (r % world != rank) continue;
            BoxResult res = minimize_box_lbfgs(P, starts + ((size_t)a * nstart + r) * nh, max_iter, tol);
            if (hip_rc != GPMPC_OK) return hip_rc;               // device failure: g_err holds the text
            // The linear noise variable is badly scaled against the log variables (its whole box is 1e-2 wide): once the
            // first search has stopped with iterations to spare, a second one from there with sn in log space -- where its
            // gradient no longer vanishes -- polishes the optimum (third reference-made fixture: -95.7 -> the -197.7 that
            // SLSQP with the analytic gradient finds; the reference's own run stops at -80.3).
            if (res.ok && res.iters < max_iter && P.lb[d + 1] > 0.0 && P.ub[d + 1] < inf) {
                BoxProblem P2 = P;
                P2.logv[d + 1] = 1;
                const BoxResult res2 = minimize_box_lbfgs(P2, res.theta.data(), max_iter - res.iters, tol);
                if (hip_rc != GPMPC_OK) return hip_rc;
                if (res2.ok && res2.f < res.f) res = res2;
            }
            std::memcpy(out + 1, res.theta.data(), nh * sizeof(double));
            if (res.ok) out[0] = res.f;
        }
    }
    if (rccl_comm) {    // one all-gather of the whole table: (1 + nh) doubles per restart (also at world = 1: a self-gather)
        RcclApi& R = rccl_api();
        if (!R.ok()) return fail(GPMPC_EHIP, "librccl.so could not be loaded");
        const size_t cnt = table.size();
        double *dsend = nullptr, *drecv = nullptr;
        HIPCHK(hipMalloc(&dsend, cnt * sizeof(double)));
        HIPCHK(hipMalloc(&drecv, cnt * world * sizeof(double)));
        HIPCHK(hipMemcpyAsync(dsend, table.data(), cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
        (void)hipGetLastError();
        const int rc = R.AllGather(dsend, drecv, cnt, RCCL_FLOAT64, rccl_comm, h->stream);
        std::vector<double> all(cnt * world);
        if (rc == 0) {
            HIPCHK(hipMemcpyAsync(all.data(), drecv, all.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
        hipFree(dsend);
        hipFree(drecv);
        if (rc != 0) return fail(GPMPC_EHIP, "ncclAllGather failed: %s", R.GetErrorString ? R.GetErrorString(rc) : "?");
        for (int a = 0; a < Ny; ++a)
            for (int r = 0; r < nstart; ++r)
                std::memcpy(&table[((size_t)a * nstart + r) * row], &all[(size_t)(r % world) * cnt + ((size_t)a * nstart + r) * row],
                            row * sizeof(double));
    }
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
    if (!merged) return GPMPC_OK;                               // caller merges the ranks' tables and calls gpmpc_fit
    if (!all_ok) return fail(GPMPC_ENOTPD, "every restart of an output failed (K not positive definite along the way)");
    return gpmpc_fit(h, hyper_opt, want_invK, info);            // optimize.py:476-494 at theta*
}

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
