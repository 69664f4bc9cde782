// hipemu.cpp -- scheduler of the TEST-ONLY HIP emulator (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#include <omp.h>

namespace hipemu {
thread_local Block* g_blk = nullptr;

static constexpr size_t STACK = 96 * 1024;

#if defined(__x86_64__) && !defined(HIPEMU_UCONTEXT)
// swapcontext() saves and restores the signal mask: two system calls per yield, and a barrier of a 256-thread workgroup is 512 yields.  The
// fibers here never touch signals, so the switch is the callee-saved registers and the stack pointer (System V x86-64 ABI).
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
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
    .size hipemu_switch,.-hipemu_switch
)");

void yield_to_sched()
{
    Block* b = g_blk;
    hipemu_switch(&b->fibers[b->cur].sp, b->sched_sp);
}

static void fiber_entry()
{
    Block* b = g_blk;
    (*b->body)();
    b->fibers[b->cur].st = DONE;
    for (;;) yield_to_sched();          // a finished fiber is never resumed; it must not return (there is no frame below it)
}

static void fiber_init(Block& b, Fiber& f)
{
    // the first switch pops six registers and `ret`s into fiber_entry with the stack as after a call (rsp = 8 mod 16)
    uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
    void** sp = (void**)(top - 64);
    for (int i = 0; i < 8; i++) sp[i] = nullptr;
    sp[6] = (void*)fiber_entry;
    f.sp = sp;
}
static inline void resume(Block& b, Fiber& f) { hipemu_switch(&b.sched_sp, f.sp); }
#else
void yield_to_sched()
{
    Block* b = g_blk;
    Fiber& f = b->fibers[b->cur];
    swapcontext(&f.ctx, &b->sched);
}

static void fiber_entry()
{
    Block* b = g_blk;
    (*b->body)();
    b->fibers[b->cur].st = DONE;
    // returning switches to uc_link (= scheduler)
}

static void fiber_init(Block& b, Fiber& f)
{
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = STACK;
    f.ctx.uc_link = &b.sched;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
}
static inline void resume(Block& b, Fiber& f) { swapcontext(&b.sched, &f.ctx); }
#endif

static void run_block(Block& b, int nthreads)
{
    g_blk = &b;
    for (int t = 0; t < nthreads; t++) {
        Fiber& f = b.fibers[t];
        f.st = READY;
        f.tid = dim3(t % b.bdim.x, (t / b.bdim.x) % b.bdim.y, t / (b.bdim.x * b.bdim.y));
        fiber_init(b, f);
    }
    b.barrier_acc = 0; b.barrier_count = 0;
    const int nwaves = (nthreads + 63) / 64;
    // The order in which READY fibres run between two rendezvous is not part of the programming model: a kernel whose result depends on it is
    // missing a barrier (or leans on lockstep execution the source does not ask for).  HIPEMU_ORDER=reverse | random[:seed] runs them in another
    // order than 0, 1, 2, ... -- a poor man's race detector for the tests (scripts/exp/emu_orders.sh).
    static const int order_mode = [] { const char* e = std::getenv("HIPEMU_ORDER"); return !e ? 0 : e[0] == 'r' && e[1] == 'e' ? 1 : e[0] == 'r' ? 2 : 0; }();
    static const unsigned order_seed = [] { const char* e = std::getenv("HIPEMU_ORDER"); const char* c = e ? std::strchr(e, ':') : nullptr; return c ? (unsigned)std::atoi(c + 1) : 1u; }();
    static thread_local std::vector<int> order;
    order.resize(nthreads);
    for (int t = 0; t < nthreads; t++) order[t] = order_mode == 1 ? nthreads - 1 - t : t;
    unsigned rng = order_seed * 2654435761u + b.bid.x * 40503u + b.bid.y * 9973u + 12345u;
    for (unsigned pass = 0;; pass++) {
        bool ran = false;
        if (order_mode == 2)                                       // a fresh permutation for every pass (Fisher-Yates, xorshift)
            for (int t = nthreads - 1; t > 0; t--) { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; std::swap(order[t], order[rng % (unsigned)(t + 1)]); }
        for (int oi = 0; oi < nthreads; oi++) {
            const int t = order[oi];
            if (b.fibers[t].st != READY) continue;
            ran = true;
            b.cur = t;
            resume(b, b.fibers[t]);
        }
        // release wave collectives: after a full pass no fiber is READY, so the lanes waiting at a
        // collective are exactly the active lanes of it.
        bool released = false;
        for (int w = 0; w < nwaves; w++) {
            unsigned long long mask = 0; int op = -1; bool mixed = false;
            for (int l = 0; l < 64 && w * 64 + l < nthreads; l++) {
                Fiber& f = b.fibers[w * 64 + l];
                if (f.st == AT_WAVE) {
                    mask |= 1ull << l;
                    if (op < 0) op = f.wave_op; else if (op != f.wave_op) mixed = true;
                }
            }
            if (!mask) continue;
            if (mixed) { std::fprintf(stderr, "hipemu: divergent wave collectives in block (%u,%u) wave %d\n", b.bid.x, b.bid.y, w); std::abort(); }
            for (int l = 0; l < 64 && w * 64 + l < nthreads; l++) {
                Fiber& f = b.fibers[w * 64 + l];
                b.wave_out[w][l] = (f.st == AT_WAVE) ? f.wave_in : 0ull;
                if (f.st == AT_WAVE) f.st = READY;
            }
            b.wave_mask[w] = mask;
            released = true;
        }
        if (released) continue;
        int nbar = 0, ndone = 0;
        for (int t = 0; t < nthreads; t++) { nbar += b.fibers[t].st == AT_BARRIER; ndone += b.fibers[t].st == DONE; }
        if (ndone == nthreads) break;
        if (nbar + ndone == nthreads && nbar > 0) {
            b.barrier_count = b.barrier_acc; b.barrier_acc = 0;
            for (int t = 0; t < nthreads; t++) if (b.fibers[t].st == AT_BARRIER) b.fibers[t].st = READY;
            continue;
        }
        if (!ran) { std::fprintf(stderr, "hipemu: deadlock in block (%u,%u)\n", b.bid.x, b.bid.y); std::abort(); }
    }
    g_blk = nullptr;
}

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body)
{
    const int nthreads = (int)(block.x * block.y * block.z);
    const long long nblocks = (long long)grid.x * grid.y * grid.z;
    if (nthreads > 1024) { std::fprintf(stderr, "hipemu: block too large\n"); std::abort(); }
#pragma omp parallel
    {
        Block b;
        b.fibers.resize(nthreads);
        // the fiber stacks of an OS thread live as long as the thread (OpenMP keeps its pool): a fresh zero-filled 24 MB vector per thread and
        // launch was most of the emulated suite's system time (page faults)
        static thread_local char* stacks = nullptr;
        static thread_local int stacks_for = 0;
        if (stacks_for < nthreads) {
            std::free(stacks);
            stacks = (char*)std::malloc((size_t)nthreads * STACK + 64);
            if (!stacks) { std::fprintf(stderr, "hipemu: out of memory for the fiber stacks\n"); std::abort(); }
            stacks_for = nthreads;
        }
        for (int t = 0; t < nthreads; t++) b.fibers[t].stack = stacks + (size_t)t * STACK;
        b.bdim = block; b.gdim = grid; b.body = &body;
#pragma omp for schedule(dynamic, 1)
        for (long long i = 0; i < nblocks; i++) {
            b.bid = dim3((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((long long)grid.x * grid.y)));
            run_block(b, nthreads);
        }
    }
}
}  // namespace hipemu
