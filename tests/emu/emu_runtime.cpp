// Fiber scheduler behind tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY (see that header).
#include <hip/hip_runtime.h>

#include <cassert>
#include <deque>
#include <map>
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
constexpr size_t kStack = 192 * 1024;
struct Wave {
    double va[2][64], vb[2][64];
    int count = 0;
    unsigned gen = 0;
};
struct Block {
    dim3 bid;
    int alive = 0, bar_count = 0;
    unsigned bar_gen = 0;
    std::vector<Wave> waves;
    std::vector<char> smem;
};
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    dim3 tid;
    int lin = 0;
    Block* blk = nullptr;
    const std::function<void()>* body = nullptr;
    dim3 kblock, kgrid;
};
std::vector<Fiber> fibers;
std::vector<Block> blocks;
Fiber* cur = nullptr;
void* sched_sp = nullptr;
const std::function<void()>* body_fn = nullptr;

void yield() { emu_swap(&cur->sp, sched_sp); }

void fiber_entry() {
    (*body_fn)();
    cur->done = true;
    --cur->blk->alive;
    emu_swap(&cur->sp, sched_sp);
    std::abort();  // a finished fiber is never resumed
}

// every lane of the wave calls this; returns after all 64 lanes have arrived
void wave_rendezvous(Wave& w) {
    const unsigned g = w.gen;
    if (++w.count == 64) {
        w.count = 0;
        ++w.gen;
    } else {
        while (w.gen == g) yield();
    }
}

void prepare_fiber(Fiber& f, int t, dim3 block, Block* b) {
    f.done = false;
    f.lin = t;
    f.blk = b;
    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    // initial frame for emu_swap: 6 callee-saved slots, then the entry address
    uintptr_t top = ((uintptr_t)f.stack + kStack - 64) & ~(uintptr_t)15;
    void** sp = (void**)top;
    sp[6] = (void*)&fiber_entry;
    for (int i = 0; i < 6; ++i) sp[i] = nullptr;
    f.sp = (void*)sp;  // sp % 16 == 0  ->  rsp % 16 == 8 at fiber_entry
}

}  // namespace

int lane() { return cur->lin & 63; }
void* dyn_smem() { return cur->blk->smem.data(); }
void yield_thread() { yield(); }

double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

void syncthreads() {
    Block& b = *cur->blk;
    const unsigned g = b.bar_gen;
    ++b.bar_count;
    while (b.bar_gen == g) {
        if (b.bar_count >= b.alive) {
            b.bar_count = 0;
            ++b.bar_gen;
            break;
        }
        yield();
    }
}

double wave_xchg(double v, int src_lane) {
    Wave& w = cur->blk->waves[cur->lin >> 6];
    const int p = (int)(w.gen & 1u);
    w.va[p][cur->lin & 63] = v;
    wave_rendezvous(w);
    return w.va[p][src_lane & 63];
}

// v_mfma_f64_16x16x4_f64 fragment layout (cdna_hip_programming.md section 3): lane l holds
// A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; result register r of lane l is
// D[row = (l>>4) + 4 r][col = l&15].
void mfma_f64_16x16x4(double a, double b, const double* c, double* d) {
    Wave& w = cur->blk->waves[cur->lin >> 6];
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

// ---------------------------------------------------------------- asynchronous stream model
namespace {
struct Op {
    int kind;  // 0 kernel, 1 record event, 2 wait event
    dim3 grid, block;
    std::function<void()> body;
    size_t shmem = 0;
    hipEvent_t ev = nullptr;
    long seq = 0;
};
struct Running {            // a kernel that is executing: its workgroup contexts and fibers
    Op op;
    std::vector<Block> blocks;
    std::vector<Fiber*> fibs;
    size_t next_block = 0;  // sequential mode: index of the workgroup that runs next
    size_t nblocks = 0;
    bool concurrent = false;
};
std::map<int, std::deque<Op>> queues;   // stream id -> FIFO
std::vector<char*> free_stacks;
bool draining = false;

Fiber* new_fiber() {
    Fiber* f = new Fiber();
    if (!free_stacks.empty()) { f->stack = free_stacks.back(); free_stacks.pop_back(); }
    else { void* s = nullptr; if (posix_memalign(&s, 64, kStack) != 0) std::abort(); f->stack = (char*)s; }
    return f;
}
void free_fiber(Fiber* f) { free_stacks.push_back(f->stack); delete f; }

dim3 block_coord(const dim3& grid, size_t i) {
    return dim3((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((size_t)grid.x * grid.y)));
}

void start_block(Running& r, size_t slot, size_t bidx) {
    const int nthreads = (int)(r.op.block.x * r.op.block.y * r.op.block.z);
    Block& b = r.blocks[slot];
    b.bid = block_coord(r.op.grid, bidx);
    b.bar_count = 0;
    b.alive = nthreads;
    for (auto& w : b.waves) w.count = 0;
    for (int t = 0; t < nthreads; ++t) {
        Fiber* f = new_fiber();
        prepare_fiber(*f, t, r.op.block, &b);
        f->body = &r.op.body;
        f->kblock = r.op.block;
        f->kgrid = r.op.grid;
        r.fibs.push_back(f);
    }
}

Running* start_kernel(const Op& op) {
    Running* r = new Running();
    r->op = op;
    const int nthreads = (int)(op.block.x * op.block.y * op.block.z);
    r->nblocks = (size_t)op.grid.x * op.grid.y * op.grid.z;
    r->concurrent = op.shmem > 0;
    r->blocks.assign(r->concurrent ? r->nblocks : 1, Block());
    for (auto& b : r->blocks) {
        b.waves.assign(nthreads / 64, Wave());
        b.smem.assign(op.shmem ? op.shmem + 64 : 0, 0);
    }
    if (r->concurrent) {
        for (size_t i = 0; i < r->nblocks; ++i) start_block(*r, i, i);
        r->next_block = r->nblocks;
    } else {
        start_block(*r, 0, 0);
        r->next_block = 1;
    }
    return r;
}

// Static __shared__ arrays are plain C++ statics here, so two workgroups that use them must never
// interleave -- not even workgroups of different kernels on different streams (same template
// instantiation = same statics).  A workgroup of a static-LDS kernel therefore holds this token from
// its first to its last instruction; dynamic-LDS (concurrent) kernels are not affected.
Running* static_lds_owner = nullptr;

// one scheduling pass over a running kernel; returns true when the kernel has finished
bool step_kernel(Running& r) {
    if (!r.concurrent) {
        if (static_lds_owner && static_lds_owner != &r) return false;
        static_lds_owner = &r;
    }
    bool any = false;
    for (size_t i = 0; i < r.fibs.size(); ++i) {
        Fiber* f = r.fibs[i];
        if (f->done) continue;
        any = true;
        cur = f;
        threadIdx = f->tid;
        blockIdx = f->blk->bid;
        blockDim = f->kblock;
        gridDim = f->kgrid;
        body_fn = f->body;
        emu_swap(&sched_sp, f->sp);
    }
    bool alldone = true;
    for (Fiber* f : r.fibs) if (!f->done) { alldone = false; break; }
    if (!alldone) return false;
    for (Fiber* f : r.fibs) free_fiber(f);
    r.fibs.clear();
    if (!r.concurrent) static_lds_owner = nullptr;
    if (!r.concurrent && r.next_block < r.nblocks) {   // next workgroup of a sequential kernel
        start_block(r, 0, r.next_block++);
        return false;
    }
    (void)any;
    return true;
}
}  // namespace

void launch(dim3 grid, dim3 block, const std::function<void()>& body, size_t shmem, hipStream_t stream) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads % 64 != 0) {
        std::fprintf(stderr, "emu: block size %d is not a multiple of the wave size 64\n", nthreads);
        std::abort();
    }
    Op op;
    op.kind = 0; op.grid = grid; op.block = block; op.body = body; op.shmem = shmem;
    queues[stream ? stream->id : 0].push_back(std::move(op));
}

void record(hipEvent_t e, hipStream_t s) {
    Op op; op.kind = 1; op.ev = e; op.seq = ++e->seq_recorded;
    queues[s ? s->id : 0].push_back(std::move(op));
}

void wait_event(hipStream_t s, hipEvent_t e) {
    Op op; op.kind = 2; op.ev = e; op.seq = e->seq_recorded;   // waits for the most recent record
    queues[s ? s->id : 0].push_back(std::move(op));
}

void drain() {
    if (draining) return;
    draining = true;
    std::map<int, Running*> active;
    long idle_passes = 0;
    for (;;) {
        bool pending = false, progressed = false;
        for (auto& kv : queues) {
            const int sid = kv.first;
            auto& q = kv.second;
            if (active.count(sid)) {
                pending = true;
                if (step_kernel(*active[sid])) { delete active[sid]; active.erase(sid); }
                progressed = true;
                continue;
            }
            while (!q.empty()) {
                Op& op = q.front();
                if (op.kind == 1) { op.ev->t = now_ms(); op.ev->seq_done = op.seq; q.pop_front(); progressed = true; continue; }
                if (op.kind == 2) {
                    if (op.ev->seq_done >= op.seq) { q.pop_front(); progressed = true; continue; }
                    pending = true;
                    break;   // blocked on an event of another stream
                }
                active[sid] = start_kernel(op);
                q.pop_front();
                pending = true;
                progressed = true;
                break;
            }
        }
        if (!pending && active.empty()) break;
        if (!progressed && ++idle_passes > 1000) { std::fprintf(stderr, "emu: stream dead-lock (event never recorded)\n"); std::abort(); }
    }
    draining = false;
}
}  // namespace emu
