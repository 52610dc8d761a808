// Where does the time of the tile-owner worker's trailing update go?  224 workgroups x 512 threads, each applying
// T rank-64 updates C(64x64) -= A(64x64) B(64x64)^T with operand blocks taken from a 2 MB panel column in memory,
// exactly as part 3 of chol_worker_kernel does (DMA-staged operand pairs, two images, one barrier per tile).
// ABL bits: 1 no DMA loads, 2 no fragment reads, 4 no barrier, 8 no MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../gp_mpc_amd/csrc/mfma_f64.hpp"
#include "../../gp_mpc_amd/csrc/lds_dma.hpp"
using namespace gpmpc;

template <int ABL, int NACC>
__global__ void __launch_bounds__(512) k(const double* __restrict__ L0, long ld, int nblk, int T, double* out, int cold) {
    const double* L = L0;
    char* smem = (char*)GPMPC_DYN_SMEM();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int drow = 8 * wave + (lane >> 3);
    const unsigned dvo = 8u * (unsigned)(drow * (int)ld) + ((unsigned)((lane & 7) ^ ((drow >> 1) & 7)) << 4);
    const unsigned fsw = (unsigned)(((lane & 15) >> 1) & 7), fq = (unsigned)(lane >> 4);
    const unsigned fa0 = (unsigned)((16 * wr + (lane & 15)) * 128) + ((fq ^ fsw) << 4), fa1 = fa0 ^ 64u;
    const unsigned fb0 = (unsigned)(32768 + (32 * wc + (lane & 15)) * 128) + ((fq ^ fsw) << 4), fb1 = fb0 ^ 64u;
    d4 C[NACC][2];
    for (int n = 0; n < NACC; ++n) C[n][0] = C[n][1] = d4{0, 0, 0, 0};
    auto request = [&](int t, int pp) {
        if (ABL & 1) return;
        char* img = smem + pp * 65536 + 1024 * wave;
        const int i = (blockIdx.x * 7 + t * 3) % nblk, j = (blockIdx.x * 5 + t) % nblk;
        const double* L = L0 + (cold ? (long)((t / 9) % 64) * 64 * nblk * ld : 0);     // a fresh panel column every 9 tiles
        const dma_rsrc_t ra = dma_make_rsrc(L + (long)(64 * i) * ld, (unsigned)(64 * ld * 8));
        const dma_rsrc_t rb = dma_make_rsrc(L + (long)(64 * j) * ld, (unsigned)(64 * ld * 8));
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) dma_load16(ra, img + 8192 * sl, dvo, 128u * sl);
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) dma_load16(rb, img + 32768 + 8192 * sl, dvo, 128u * sl);
    };
    request(0, 0);
    int pp = 0;
    for (int t0 = 0; t0 < T; t0 += NACC) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            const int t = t0 + n;
            if (cold == 2 && n == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            dma_wait<0>();
            if (!(ABL & 4)) dma_barrier();
            if (!(ABL & 16) && !(ABL & 32) && t + 1 < T) request(t + 1, pp ^ 1);
            double2 stage[8];
            const bool rs = (ABL & 32) && t + 1 < T;
            if (rs) {      // register staging: this thread's 8 pieces of 16 bytes of the next pair (piece = wave-load index)
                const int i = (blockIdx.x * 7 + (t + 1) * 3) % nblk, j = (blockIdx.x * 5 + t + 1) % nblk;
                const double* Ln = L0 + (cold ? (long)(((t + 1) / 9) % 64) * 64 * nblk * ld : 0);
                const char* pa = (const char*)(Ln + (long)(64 * i) * ld);
                const char* pb = (const char*)(Ln + (long)(64 * j) * ld);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    stage[q] = *reinterpret_cast<const double2*>((q < 4 ? pa : pb) + dvo + 128u * (q & 3));
            }
            const char* img = smem + pp * 65536;
            dma_rsrc_t nra, nrb;
            const bool more = (ABL & 16) && t + 1 < T;
            if (more) {
                const int i = (blockIdx.x * 7 + (t + 1) * 3) % nblk, j = (blockIdx.x * 5 + t + 1) % nblk;
                const double* Ln = L0 + (cold ? (long)(((t + 1) / 9) % 64) * 64 * nblk * ld : 0);
                nra = dma_make_rsrc(Ln + (long)(64 * i) * ld, (unsigned)(64 * ld * 8));
                nrb = dma_make_rsrc(Ln + (long)(64 * j) * ld, (unsigned)(64 * ld * 8));
            }
#pragma unroll 1
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (more) {                      // piece q = 2 sl + h of the next pair: one DMA per four MFMAs
                        const int q = 2 * sl + h;
                        dma_load16_relaxed(q < 4 ? nra : nrb, smem + (pp ^ 1) * 65536 + (q >> 2) * 32768 + 8192 * (q & 3) + 1024 * wave, dvo, 128u * (q & 3));
                    }
                    double2 a, b0, b1;
                    if (ABL & 2) { a = double2{1.0 + lane, 2.0}; b0 = double2{3.0, 1.0 + sl}; b1 = double2{0.5 * h, 1.5}; }
                    else {
                        a = *reinterpret_cast<const double2*>(img + 8192 * sl + (h ? fa1 : fa0));
                        b0 = *reinterpret_cast<const double2*>(img + 8192 * sl + (h ? fb1 : fb0));
                        b1 = *reinterpret_cast<const double2*>(img + 8192 * sl + 2048 + (h ? fb1 : fb0));
                    }
                    a.x = -a.x; a.y = -a.y;
                    if (!(ABL & 8)) {
                        C[n][0] = mfma16(a.x, b0.x, C[n][0]);
                        C[n][1] = mfma16(a.x, b1.x, C[n][1]);
                        C[n][0] = mfma16(a.y, b0.y, C[n][0]);
                        C[n][1] = mfma16(a.y, b1.y, C[n][1]);
                    } else { C[n][0][0] += a.x * b0.x; C[n][1][0] += a.y * b1.y; }
                }
            if (rs) {
                char* img = smem + (pp ^ 1) * 65536 + 1024 * wave + 16 * lane;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    *reinterpret_cast<double2*>(img + (q >> 2) * 32768 + 8192 * (q & 3)) = stage[q];
            }
            pp ^= 1;
        }
    }
    double s = 0;
    for (int n = 0; n < NACC; ++n) s += C[n][0][0] + C[n][1][3];
    out[(long)blockIdx.x * 512 + tid] = s;
}

template <int ABL, int NACC>
void run(const char* name, const double* L, long ld, int nblk, double* out, int nwg, int cold = 0) {
    const int T = 9 * 64;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<ABL, NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<ABL, NACC>), dim3(nwg), dim3(512), 131072 + 256, 0, L, ld, nblk, 9, out, cold);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<ABL, NACC>), dim3(nwg), dim3(512), 131072 + 256, 0, L, ld, nblk, T, out, cold);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  %-52s %3d WGs: %7.3f us per tile  (%.1f TFLOP/s over the launch)\n", name, nwg, ms * 1e3 / T, 2.0 * 64 * 64 * 64 * T * nwg / (ms * 1e-3) * 1e-12);
}

int main() {
    const long ld = 4096; const int nblk = 63;
    double *L, *out;
    hipMalloc(&L, (size_t)64 * 64 * nblk * ld * 8); hipMalloc(&out, 256 * 512 * 8);    // 64 panel columns of 132 MB (ld = N)
    hipMemset(L, 0, (size_t)64 * 64 * nblk * ld * 8);
    run<32, 9>("register-staged (16-byte loads, ds_write_b128)", L, ld, nblk, out, 224, 0);
    run<32, 9>("register-staged, fresh column + acquire fence", L, ld, nblk, out, 224, 2);
    run<16, 9>("interleaved requests", L, ld, nblk, out, 224, 0);
    run<16, 9>("interleaved requests, fresh column + acquire fence", L, ld, nblk, out, 224, 2);
    run<0, 9>("full, fresh panel column every 9 tiles", L, ld, nblk, out, 224, 1);
    run<0, 9>("full, fresh column + acquire fence", L, ld, nblk, out, 224, 2);
    run<8, 9>("no MFMAs, fresh column + acquire fence", L, ld, nblk, out, 224, 2);
    for (int nwg : {224}) {
        run<0, 9>("full", L, ld, nblk, out, nwg);
        run<1, 9>("no DMA loads", L, ld, nblk, out, nwg);
        run<2, 9>("no fragment reads", L, ld, nblk, out, nwg);
        run<3, 9>("no loads, no fragment reads", L, ld, nblk, out, nwg);
        run<7, 9>("MFMAs only", L, ld, nblk, out, nwg);
        run<8, 9>("no MFMAs", L, ld, nblk, out, nwg);
        run<4, 9>("no barrier", L, ld, nblk, out, nwg);
    }
    return 0;
}
