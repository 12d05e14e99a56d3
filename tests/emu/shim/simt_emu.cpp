// TEST INFRASTRUCTURE -- see simt_emu.h
#include "simt_emu.h"

#include <sys/mman.h>
#include <ucontext.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace simt {
namespace {

constexpr size_t kStack = 4u << 20;   // virtual, committed lazily (the unrolled MLP kernels have big frames)

struct Wave {
    int gen = 0, arrived = 0, active = 0;
    alignas(64) uint64_t xbuf[2][64];
};

struct Block;
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    int lane = 0;
    Wave* wave = nullptr;
    unsigned parity = 0;
    const int* wait_gen = nullptr;
    int wait_val = 0;
    Idx3 tidx{0, 0, 0};
};

struct Block {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    int gen = 0, arrived = 0, active = 0;
    char* lds = nullptr;
    Idx3 bidx, bdim, gdim;
    ucontext_t sched;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
};

thread_local Block* tl_block = nullptr;
thread_local std::vector<char*>* tl_stacks = nullptr;
int g_threads = 0;

void yield_to_sched() {
    Block* b = tl_block;
    Fiber* f = b->cur;
    swapcontext(&f->ctx, &b->sched);
}

void release_wave_if_complete(Wave* w) {
    if (w->active > 0 && w->arrived >= w->active) { w->arrived = 0; w->gen++; }
}
void release_block_if_complete(Block* b) {
    if (b->active > 0 && b->arrived >= b->active) { b->arrived = 0; b->gen++; }
}

void trampoline() {
    Block* b = tl_block;
    Fiber* f = b->cur;
    (*b->body)();
    f->done = true;
    // a thread that has returned no longer takes part in rendezvous
    f->wave->active--;
    release_wave_if_complete(f->wave);
    b->active--;
    release_block_if_complete(b);
    swapcontext(&f->ctx, &b->sched);
    abort();
}

void run_block(Block& b) {
    tl_block = &b;
    if (!tl_stacks) tl_stacks = new std::vector<char*>();
    const unsigned nthreads = b.bdim.x * b.bdim.y * b.bdim.z;
    while (tl_stacks->size() < nthreads) {
        void* p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { perror("mmap"); abort(); }
        tl_stacks->push_back((char*)p);
    }
    b.fibers.assign(nthreads, Fiber());
    b.waves.assign((nthreads + 63) / 64, Wave());
    b.active = (int)nthreads;
    for (unsigned t = 0; t < nthreads; ++t) {
        Fiber& f = b.fibers[t];
        f.stack = (*tl_stacks)[t];
        f.lane = (int)(t & 63);
        f.wave = &b.waves[t >> 6];
        f.wave->active++;
        f.tidx = {t % b.bdim.x, (t / b.bdim.x) % b.bdim.y, t / (b.bdim.x * b.bdim.y)};
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    unsigned remaining = nthreads;
    while (remaining) {
        bool progress = false;
        for (unsigned t = 0; t < nthreads; ++t) {
            Fiber& f = b.fibers[t];
            if (f.done) continue;
            if (f.wait_gen) {
                if (*f.wait_gen == f.wait_val) continue;
                f.wait_gen = nullptr;
            }
            b.cur = &f;
            swapcontext(&b.sched, &f.ctx);
            progress = true;
            if (f.done) remaining--;
        }
        if (!progress) {
            fprintf(stderr, "[simt_emu] deadlock in block (%u,%u,%u): every live thread waits at a "
                            "barrier that cannot complete (mismatched __syncthreads / wave op?)\n",
                    b.bidx.x, b.bidx.y, b.bidx.z);
            abort();
        }
    }
    tl_block = nullptr;
}

}  // namespace

const Idx3& cur_thread_idx() { return tl_block->cur->tidx; }
const Idx3& cur_block_idx() { return tl_block->bidx; }
const Idx3& cur_block_dim() { return tl_block->bdim; }
const Idx3& cur_grid_dim() { return tl_block->gdim; }
int cur_lane() { return tl_block->cur->lane; }
char* block_lds() { return tl_block->lds; }

const uint64_t* wave_exchange(uint64_t v) {
    Fiber* f = tl_block->cur;
    Wave* w = f->wave;
    const unsigned slot = (f->parity++) & 1u;
    w->xbuf[slot][f->lane] = v;
    const int g = w->gen;
    w->arrived++;
    if (w->arrived >= w->active) {
        w->arrived = 0;
        w->gen = g + 1;
    } else {
        f->wait_gen = &w->gen;
        f->wait_val = g;
        yield_to_sched();
    }
    return w->xbuf[slot];
}

void block_barrier() {
    Block* b = tl_block;
    Fiber* f = b->cur;
    const int g = b->gen;
    b->arrived++;
    if (b->arrived >= b->active) {
        b->arrived = 0;
        b->gen = g + 1;
    } else {
        f->wait_gen = &b->gen;
        f->wait_val = g;
        yield_to_sched();
    }
}

void set_threads(int n) { g_threads = n; }

void launch(Idx3 grid, Idx3 block, size_t lds_bytes, const std::function<void()>& body) {
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    int nt = g_threads;
    if (nt <= 0) {
        const char* e = getenv("SIMT_EMU_THREADS");
        nt = e ? atoi(e) : 8;
        if (nt <= 0) nt = 1;
    }
    if ((size_t)nt > nblocks) nt = (int)nblocks;
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        std::vector<char> lds(lds_bytes + 64);
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= nblocks) break;
            Block b;
            b.bidx = {(unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((size_t)grid.x * grid.y))};
            b.bdim = block;
            b.gdim = grid;
            char* p = lds.data();
            p += (64 - ((uintptr_t)p & 63)) & 63;
            memset(p, 0xCD, lds_bytes);   // poison: uninitialised LDS reads show up
            b.lds = p;
            b.body = &body;
            run_block(b);
        }
    };
    if (nt == 1) {
        worker();
    } else {
        std::vector<std::thread> ts;
        for (int t = 0; t < nt; ++t) ts.emplace_back(worker);
        for (auto& t : ts) t.join();
    }
}

}  // namespace simt
