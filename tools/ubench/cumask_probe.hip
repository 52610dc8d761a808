// cumask_probe: does hipExtStreamCreateWithCUMask work for an ordinary user on this box, and which physical CUs does bit i of
// the mask select?  Every workgroup records (XCC_ID, SE, SH, CU) from the hardware-id registers and then spins for a while so
// that the launch has to spread over every CU the queue may use.  Also times two MFMA-free spin kernels on complementary masks
// (do they run side by side?) and an unmasked launch next to a masked one.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/cumask_probe.hip -o tools/ubench/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#include <map>

__global__ void probe(unsigned* out, long spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static int run(hipStream_t st, const char* name, int wgs, int threads, long spin, size_t lds) {
    unsigned* d; CK(hipMalloc(&d, wgs * 8));
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(threads), lds, st, d, spin);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned> h(2 * wgs); CK(hipMemcpy(h.data(), d, wgs * 8, hipMemcpyDeviceToHost));
    std::map<int, std::set<int>> per_xcc;   // xcc -> set of (se, sh, cu)
    std::map<int, int> wg_xcc;
    for (int i = 0; i < wgs; ++i) {
        unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
        int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc].insert(se * 32 + sh * 16 + cu);
        wg_xcc[i] = xcc;
    }
    int total = 0;
    printf("%s: %d workgroups\n", name, wgs);
    for (auto& kv : per_xcc) {
        printf("  xcc %d: %2zu CUs:", kv.first, kv.second.size());
        for (int c : kv.second) printf(" %d.%d.%d", c / 32, (c / 16) & 1, c & 15);
        printf("\n");
        total += (int)kv.second.size();
    }
    printf("  distinct CUs %d; first 16 workgroups' xcc:", total);
    for (int i = 0; i < 16 && i < wgs; ++i) printf(" %d", wg_xcc[i]);
    printf("\n");
    hipFree(d);
    return 0;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    const long spin = 20000;    // wall_clock64 ticks at 100 MHz: 200 us
    hipStream_t s0; CK(hipStreamCreate(&s0));
    run(s0, "no mask", 1024, 64, spin, 0);
    struct M { const char* name; std::vector<uint32_t> m; };
    std::vector<M> masks;
    { std::vector<uint32_t> m(8, 0); m[0] = 0xffffffffu; masks.push_back({"bits 0-31", m}); }
    { std::vector<uint32_t> m(8, 0); m[0] = 0xff; masks.push_back({"bits 0-7", m}); }
    { std::vector<uint32_t> m(8, 0); m[0] = 0x1; masks.push_back({"bit 0", m}); }
    { std::vector<uint32_t> m(8, 0); m[0] = 0x100; masks.push_back({"bit 8", m}); }
    { std::vector<uint32_t> m(8, 0); m[1] = 0xffffffffu; masks.push_back({"bits 32-63", m}); }
    { std::vector<uint32_t> m(8, 0xffffffffu); m[0] = 0; masks.push_back({"bits 32-255", m}); }
    { std::vector<uint32_t> m(8, 0xffffffffu); m[7] = 0; masks.push_back({"bits 0-223", m}); }
    { std::vector<uint32_t> m(8, 0); m[7] = 0xffffffffu; masks.push_back({"bits 224-255", m}); }
    std::vector<hipStream_t> st(masks.size());
    for (size_t i = 0; i < masks.size(); ++i) {
        hipError_t e = hipExtStreamCreateWithCUMask(&st[i], (uint32_t)masks[i].m.size(), masks[i].m.data());
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask(%s) -> %s\n", masks[i].name, hipGetErrorString(e)); return 2; }
        run(st[i], masks[i].name, 1024, 64, spin, 0);
    }
    // a workgroup that needs a whole CU's LDS (like the chain kernel) on the one-CU mask
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    run(st[2], "bit 0, 150 KB LDS workgroups", 4, 256, spin, 150 * 1024);
    // concurrency: masked (bits 0-223) and masked (bits 224-255) side by side
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned *da, *db; hipMalloc(&da, 8 * 4096); hipMalloc(&db, 8 * 4096);
    for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        hipEventRecord(e0, s0);
        hipStreamWaitEvent(st[6], e0, 0); hipStreamWaitEvent(st[7], e0, 0);
        hipLaunchKernelGGL(probe, dim3(448), dim3(512), 64 * 1024, st[6], da, 100000L);    // 1 ms, 2 per CU by LDS
        hipLaunchKernelGGL(probe, dim3(64), dim3(512), 64 * 1024, st[7], db, 100000L);
        hipEventRecord(e1, st[6]); hipStreamWaitEvent(s0, e1, 0);
        hipEventRecord(e1, st[7]); hipStreamWaitEvent(s0, e1, 0);
        hipEventRecord(e1, s0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("448 + 64 one-ms workgroups (64 KB LDS each: two per CU) on complementary masks: %.3f ms (1.0 = side by side, all resident)\n", ms);
    }
    return 0;
}
