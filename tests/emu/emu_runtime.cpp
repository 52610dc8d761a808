// Fiber scheduler behind tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY (see that header).
#include <hip/hip_runtime.h>

#include <cassert>
#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

// x86-64 SysV context switch: save callee-saved registers + stack pointer, load the other's.
extern "C" void emu_swap(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_swap
.type emu_swap,@function
emu_swap:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_swap,.-emu_swap
)");

namespace emu {
namespace {
constexpr size_t kStack = 256 * 1024;
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    dim3 tid;
    int lin = 0;
};
struct Wave {
    double va[2][64], vb[2][64];
    int count = 0;
    unsigned gen = 0;
};
std::vector<Fiber> fibers;
std::vector<Wave> waves;
Fiber* cur = nullptr;
void* sched_sp = nullptr;
const std::function<void()>* body_fn = nullptr;
int alive = 0, bar_count = 0;
unsigned bar_gen = 0;

void yield() { emu_swap(&cur->sp, sched_sp); }

void fiber_entry() {
    (*body_fn)();
    cur->done = true;
    --alive;
    emu_swap(&cur->sp, sched_sp);
    std::abort();  // a finished fiber is never resumed
}

// every lane of the wave calls this; returns the parity buffer index that is now complete
int wave_rendezvous(Wave& w) {
    const unsigned g = w.gen;
    if (++w.count == 64) {
        w.count = 0;
        ++w.gen;
    } else {
        while (w.gen == g) yield();
    }
    return (int)(g & 1u);
}
}  // namespace

int lane() { return cur->lin & 63; }

double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

void syncthreads() {
    const unsigned g = bar_gen;
    ++bar_count;
    while (bar_gen == g) {
        if (bar_count >= alive) {
            bar_count = 0;
            ++bar_gen;
            break;
        }
        yield();
    }
}

double wave_xchg(double v, int src_lane) {
    Wave& w = waves[cur->lin >> 6];
    const int p = (int)(w.gen & 1u);
    w.va[p][cur->lin & 63] = v;
    wave_rendezvous(w);
    return w.va[p][src_lane & 63];
}

// v_mfma_f64_16x16x4_f64 fragment layout (cdna_hip_programming.md section 3): lane l holds
// A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; result register r of lane l is
// D[row = (l>>4) + 4 r][col = l&15].
void mfma_f64_16x16x4(double a, double b, const double* c, double* d) {
    Wave& w = waves[cur->lin >> 6];
    const int p = (int)(w.gen & 1u);
    const int l = cur->lin & 63;
    w.va[p][l] = a;
    w.vb[p][l] = b;
    wave_rendezvous(w);
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) + 4 * r;
        double s = c[r];
        for (int k = 0; k < 4; ++k) s = std::fma(w.va[p][row + 16 * k], w.vb[p][col + 16 * k], s);
        d[r] = s;
    }
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads % 64 != 0) {
        std::fprintf(stderr, "emu: block size %d is not a multiple of the wave size 64\n", nthreads);
        std::abort();
    }
    if ((int)fibers.size() < nthreads) {
        const size_t old = fibers.size();
        fibers.resize(nthreads);
        for (size_t i = old; i < fibers.size(); ++i) {
            void* s = nullptr;
            if (posix_memalign(&s, 64, kStack) != 0) std::abort();
            fibers[i].stack = (char*)s;
        }
    }
    waves.assign(nthreads / 64, Wave());
    body_fn = &body;
    blockDim = block;
    gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = dim3(bx, by, bz);
                bar_count = 0;
                alive = nthreads;
                for (auto& w : waves) w.count = 0;
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = fibers[t];
                    f.done = false;
                    f.lin = t;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    // initial frame for emu_swap: 6 callee-saved slots, then the entry address
                    uintptr_t top = ((uintptr_t)f.stack + kStack - 64) & ~(uintptr_t)15;
                    void** sp = (void**)top;
                    sp[6] = (void*)&fiber_entry;
                    for (int i = 0; i < 6; ++i) sp[i] = nullptr;
                    f.sp = (void*)sp;  // sp % 16 == 0  ->  rsp % 16 == 8 at fiber_entry
                }
                int remaining = nthreads;
                while (remaining > 0) {
                    const int before = remaining;
                    bool progressed = false;
                    for (int t = 0; t < nthreads; ++t) {
                        Fiber& f = fibers[t];
                        if (f.done) continue;
                        cur = &f;
                        threadIdx = f.tid;
                        emu_swap(&sched_sp, f.sp);
                        if (f.done) { --remaining; progressed = true; }
                    }
                    (void)before; (void)progressed;
                }
            }
    body_fn = nullptr;
}
}  // namespace emu
