// hipemu scheduler: see shim/hip/hip_runtime.h.  Test infrastructure only.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <vector>

namespace hipemu {
Fiber* cur = nullptr;
dim3 cur_block, cur_bdim, cur_gdim;

namespace {
constexpr size_t STACK = 256 * 1024;
constexpr unsigned MAXT = 1024;
char* g_stacks = nullptr;
void* g_sched_sp = nullptr;
std::vector<Fiber> g_fib;
std::vector<Wave> g_wave;
unsigned g_blk_gen = 0, g_blk_arrived = 0, g_blk_live = 0;
void (*g_fn)(void*) = nullptr;
void* g_arg = nullptr;
uint64_t g_rng = 0;
bool g_shuffle = false;

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
extern "C" void hipemu_set_shuffle(unsigned long long seed);
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

[[noreturn]] void die(const char* what, int a = 0, int b = 0) {
    fprintf(stderr, "hipemu: %s (%d, %d) block %u thread %u\n", what, a, b, cur_block.x, cur ? cur->tid.x : 0u);
    abort();
}

inline void yield() { hipemu_switch(&cur->sp, g_sched_sp); }

void wave_complete(Wave* w) {
    w->arrived = 0;
    w->gen++;
    WaveBuf& nb = w->buf[w->gen & 1];
    nb.present = 0;
}
void block_complete() {
    g_blk_arrived = 0;
    g_blk_gen++;
}

void fiber_main() {
    g_fn(g_arg);
    Fiber* f = cur;
    f->status = 3;
    Wave* w = f->wave;
    w->live--;
    if (w->arrived && w->arrived == w->live) wave_complete(w);
    g_blk_live--;
    if (g_blk_arrived && g_blk_arrived == g_blk_live) block_complete();
    yield();
    die("resumed a finished fiber");
}

void init_fiber(Fiber& f) {
    void** sp = (void**)(f.stack + STACK);
    *--sp = nullptr;               // keeps the entry frame 16-byte aligned (rsp % 16 == 8 at entry)
    *--sp = (void*)&fiber_main;    // popped by `ret`
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    f.sp = sp;
}

inline bool runnable(const Fiber& f) {
    switch (f.status) {
        case 0: return true;
        case 1: return f.wave->gen != f.wait_gen;
        case 2: return g_blk_gen != f.wait_gen;
        default: return false;
    }
}
}  // namespace

}  // namespace hipemu

// seed != 0: permute the order in which the lanes of a wave are resumed (race shaking); 0: lane order
extern "C" void hipemu_set_shuffle(unsigned long long seed) {
    hipemu::g_shuffle = seed != 0;
    hipemu::g_rng = seed * 0x9E3779B97F4A7C15ull + 1;
}

namespace hipemu {
const WaveBuf& wave_sync(int site, uint64_t v) {
    Fiber* f = cur;
    Wave* w = f->wave;
    WaveBuf& b = w->buf[w->gen & 1];
    if (w->arrived == 0)
        b.site = site;
    else if (b.site != site)
        die("divergent wave collective: lanes arrived from different call sites", b.site, site);
    b.vals[f->lane] = v;
    b.present |= 1ull << f->lane;
    if (++w->arrived == w->live) {
        wave_complete(w);
    } else {
        f->status = 1;
        f->wait_gen = w->gen;
        yield();
        f->status = 0;
    }
    return b;
}

void block_sync() {
    Fiber* f = cur;
    if (++g_blk_arrived == g_blk_live) {
        block_complete();
    } else {
        f->status = 2;
        f->wait_gen = g_blk_gen;
        yield();
        f->status = 0;
    }
}

void run_grid(dim3 grid, dim3 block, void (*fn)(void*), void* arg) {
    const unsigned nt = block.x * block.y * block.z;
    if (nt == 0 || nt > MAXT) die("unsupported block size", (int)nt);
    if (cur) die("nested launch");
    if (!g_stacks) {
        g_stacks = (char*)mmap(nullptr, STACK * MAXT, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == (char*)MAP_FAILED) die("mmap of fiber stacks failed");
        if (const char* e = getenv("HIPEMU_SHUFFLE")) hipemu_set_shuffle(strtoull(e, nullptr, 0));
    }
    const unsigned nw = (nt + 63) / 64;
    g_fib.resize(nt);
    g_wave.resize(nw);
    g_fn = fn;
    g_arg = arg;
    cur_bdim = block;
    cur_gdim = grid;
    unsigned order[64];
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    for (size_t bi = 0; bi < nblocks; ++bi) {
        cur_block = dim3((unsigned)(bi % grid.x), (unsigned)((bi / grid.x) % grid.y), (unsigned)(bi / ((size_t)grid.x * grid.y)));
        for (unsigned w = 0; w < nw; ++w) {
            Wave& W = g_wave[w];
            W.gen = 0;
            W.arrived = 0;
            W.live = (w + 1 < nw) ? 64 : nt - 64 * w;
            W.buf[0].present = W.buf[1].present = 0;
        }
        for (unsigned t = 0; t < nt; ++t) {
            Fiber& f = g_fib[t];
            f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f.lane = t & 63;
            f.wave = &g_wave[t >> 6];
            f.status = 0;
            f.stack = g_stacks + (size_t)t * STACK;
            init_fiber(f);
        }
        g_blk_gen = g_blk_arrived = 0;
        g_blk_live = nt;
        while (g_blk_live) {
            bool progress = false;
            for (unsigned w = 0; w < nw; ++w) {
                const unsigned base = 64 * w, n = (w + 1 < nw) ? 64 : nt - base;
                for (unsigned i = 0; i < n; ++i) order[i] = i;
                bool again = true;
                while (again) {
                    again = false;
                    if (g_shuffle)
                        for (unsigned i = n - 1; i > 0; --i) {
                            g_rng = g_rng * 6364136223846793005ull + 1442695040888963407ull;
                            unsigned j = (unsigned)((g_rng >> 33) % (i + 1));
                            unsigned tmp = order[i];
                            order[i] = order[j];
                            order[j] = tmp;
                        }
                    for (unsigned i = 0; i < n; ++i) {
                        Fiber& f = g_fib[base + order[i]];
                        if (!runnable(f)) continue;
                        cur = &f;
                        hipemu_switch(&g_sched_sp, f.sp);
                        again = progress = true;
                    }
                }
            }
            if (!progress) die("deadlock: no thread of the block can make progress (divergent barrier?)");
        }
        cur = nullptr;
    }
}
}  // namespace hipemu
