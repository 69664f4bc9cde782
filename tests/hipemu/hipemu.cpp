// hipemu.cpp -- scheduler of the TEST-ONLY HIP emulator (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#include <omp.h>

namespace hipemu {
thread_local Block* g_blk = nullptr;

static constexpr size_t STACK = 96 * 1024;

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

static void run_block(Block& b, int nthreads)
{
    g_blk = &b;
    for (int t = 0; t < nthreads; t++) {
        Fiber& f = b.fibers[t];
        f.st = READY;
        f.tid = dim3(t % b.bdim.x, (t / b.bdim.x) % b.bdim.y, t / (b.bdim.x * b.bdim.y));
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = &b.sched;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    b.barrier_acc = 0; b.barrier_count = 0;
    const int nwaves = (nthreads + 63) / 64;
    for (;;) {
        bool ran = false;
        for (int t = 0; t < nthreads; t++) {
            if (b.fibers[t].st != READY) continue;
            ran = true;
            b.cur = t;
            swapcontext(&b.sched, &b.fibers[t].ctx);
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
        std::vector<char> stacks((size_t)nthreads * STACK);
        for (int t = 0; t < nthreads; t++) b.fibers[t].stack = stacks.data() + (size_t)t * STACK;
        b.bdim = block; b.gdim = grid; b.body = &body;
#pragma omp for schedule(dynamic, 1)
        for (long long i = 0; i < nblocks; i++) {
            b.bid = dim3((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((long long)grid.x * grid.y)));
            run_block(b, nthreads);
        }
    }
}
}  // namespace hipemu
