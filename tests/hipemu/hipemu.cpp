// TEST INFRASTRUCTURE (see include/hip/hip_runtime.h): the fiber scheduler behind the host-side HIP execution model.
// One fiber per HIP thread; the fibers of a workgroup run round-robin on one OS thread and switch only at synchronisation
// points.  Workgroups are handed to a pool of OS threads (one workgroup at a time each; `static thread_local` is the
// workgroup's static LDS).  hipemu_set_coresident(1): one OS thread per workgroup, all alive at once (grid-wide waits).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <string>
#include <thread>
#include <cstring>
#include <vector>
#include <sys/mman.h>

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

namespace hipemu {

static const size_t STACK_BYTES = 256 * 1024;

struct Fiber {
    Ctx ctx;
    void* sp;
    bool done;
};
struct Wave {
    int live, arrived;
    unsigned gen;
    alignas(16) uint32_t slots[2][64 * 8];
};
struct Worker {  // per OS thread
    unsigned long wg_serial = 0;   // workgroups this OS thread has run (poison_lds: static LDS is poisoned once per workgroup)
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    char* stacks = nullptr;
    size_t nstacks = 0;
    std::vector<char> dyn;
    void* sched_sp = nullptr;
    Fiber* running = nullptr;
    int live = 0, bar_arrived = 0;
    unsigned bar_gen = 0;
    unsigned long progress = 0;
    const std::function<void()>* body = nullptr;
    ~Worker() {
        if (stacks) munmap(stacks, nstacks * STACK_BYTES);
    }
};

thread_local Ctx* cur = nullptr;
static thread_local Worker* wk = nullptr;

static std::atomic<int> g_error{0};
static std::mutex g_msg_mu;
static std::string g_msg;
static std::atomic<int> g_coresident{0};
static std::atomic<int> g_threads{0};
static std::atomic<bool> g_abort{false};
static std::atomic<bool> g_cores_now{false};  // the launch in flight runs one OS thread per workgroup
// Wave schedule of a workgroup's scheduler pass: 0 = waves in index order, 1 = reversed, 2 = a fresh pseudo-random permutation in
// every pass (seeded).  Lanes of a wave always run in lane order.  Two waves that touch the same LDS / global word between the same
// pair of barriers (a missing __syncthreads: the one thing a switch-at-sync-points model cannot see in ONE order) compute different
// results under different orders; tests/test_kernels_emu_cpu.py::test_results_do_not_depend_on_the_wave_schedule compares them bit for bit.
static std::atomic<int> g_sched_mode{0};
static std::atomic<unsigned> g_sched_seed{1};

void note_error(const char* what) {
    std::lock_guard<std::mutex> lk(g_msg_mu);
    if (g_error.exchange(hipErrorLaunchFailure) == 0) g_msg = what;
}
int last_error() { return g_error.exchange(0); }

void yield() {
    Worker* w = wk;
    Fiber* f = w->running;
    hipemu_switch(&f->sp, w->sched_sp);
}

void sleep_hint() {
    yield();
    if (g_cores_now.load(std::memory_order_relaxed)) std::this_thread::yield();
}

static void fiber_finish() {
    Worker* w = wk;
    Fiber* f = w->running;
    f->done = true;
    w->live--;
    w->waves[f->ctx.wave].live--;
    w->progress++;
    hipemu_switch(&f->sp, w->sched_sp);
    __builtin_trap();  // a finished fiber is never resumed
}

static void fiber_entry() {
    (*wk->body)();
    waitcnt_vm(0);
    fiber_finish();
}

void waitcnt_vm(int n) {
    Ctx* c = cur;
    while (c->dma_n > n) {
        const DmaOp& op = c->dma[c->dma_head];
        if (op.src)
            memcpy(op.dst, op.src, op.size);
        else
            memset(op.dst, 0, op.size);
        c->dma_head = (c->dma_head + 1) & 63;
        c->dma_n--;
    }
}

void syncthreads() {
    waitcnt_vm(0);   // what the compiler emits in front of s_barrier when an LDS-DMA may be outstanding
    barrier_raw();
}

void barrier_raw() {
    Worker* w = wk;
    const unsigned gen = w->bar_gen;
    w->bar_arrived++;
    // released by the last LIVE thread to arrive (threads that returned do not take part, as on the hardware) - the check
    // runs in every pass of the wait loop because a thread may exit while others wait
    for (;;) {
        if (w->bar_gen != gen) return;
        if (w->bar_arrived >= w->live) {
            w->bar_arrived = 0;
            w->bar_gen++;
            w->progress++;
            return;
        }
        if (g_abort.load(std::memory_order_relaxed)) fiber_finish();
        yield();
    }
}

const uint32_t* wave_exchange(const uint32_t* mine, int words) {
    Worker* w = wk;
    Ctx* c = cur;
    Wave& wv = w->waves[c->wave];
    const int ph = c->phase;
    c->phase ^= 1;
    uint32_t* slot = wv.slots[ph] + c->lane * 8;
    for (int i = 0; i < words; ++i) slot[i] = mine[i];
    const unsigned gen = wv.gen;
    wv.arrived++;
    for (;;) {
        if (wv.gen != gen) break;
        if (wv.arrived >= wv.live) {
            wv.arrived = 0;
            wv.gen++;
            w->progress++;
            break;
        }
        if (g_abort.load(std::memory_order_relaxed)) fiber_finish();
        yield();
    }
    return wv.slots[ph];
}

void* dyn_lds() { return wk->dyn.data(); }

// LDS comes up with whatever the previous workgroup left in it; the model's `static thread_local` arrays would come up ZERO on first use and a
// kernel that reads a word it never wrote (a K-tail lane, a padding row) would pass here and multiply garbage on the hardware.  With
// HIPEMU_POISON_LDS=1 (default on) every static LDS array is filled with 0xFF bytes - float / double NaN, int -1 - by the first thread of each
// workgroup that reaches its declaration (build_emu.py appends the call to the declaration), the dynamic LDS by run_block.
static const bool poison_on = !(getenv("HIPEMU_POISON_LDS") && atoi(getenv("HIPEMU_POISON_LDS")) == 0);
void poison_lds(void* p, size_t n, unsigned long* seen) {
    if (!poison_on || *seen == wk->wg_serial) return;
    *seen = wk->wg_serial;
    memset(p, 0xFF, n);
}

static void run_block(Worker& w, U3 grid, U3 block, U3 bid, size_t lds) {
    const size_t nt = (size_t)block.x * block.y * block.z;
    if (w.nstacks < nt) {
        if (w.stacks) munmap(w.stacks, w.nstacks * STACK_BYTES);
        w.stacks = (char*)mmap(nullptr, nt * STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (w.stacks == MAP_FAILED) {
            w.stacks = nullptr;
            w.nstacks = 0;
            note_error("hipemu: cannot map fiber stacks");
            return;
        }
        w.nstacks = nt;
    }
    if (w.dyn.size() < lds + 64) w.dyn.resize(lds + 64);
    w.wg_serial += 1;
    if (poison_on && lds) memset(w.dyn.data(), 0xFF, lds);
    w.fibers.assign(nt, Fiber{});
    const size_t nw = (nt + 63) / 64;
    w.waves.resize(nw);
    for (size_t i = 0; i < nw; ++i) {
        w.waves[i].live = 0;
        w.waves[i].arrived = 0;
        w.waves[i].gen = 0;
    }
    for (size_t t = 0; t < nt; ++t) {
        Fiber& f = w.fibers[t];
        f.ctx.tid = U3{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / ((size_t)block.x * block.y))};
        f.ctx.bid = bid;
        f.ctx.bdim = block;
        f.ctx.gdim = grid;
        f.ctx.linear = (int)t;
        f.ctx.lane = (int)(t & 63);
        f.ctx.wave = (int)(t >> 6);
        f.ctx.phase = 0;
        f.ctx.dma_head = f.ctx.dma_n = 0;
        f.done = false;
        w.waves[f.ctx.wave].live++;
        // initial frame: six callee-saved registers, the entry point as return address, one pad slot (ABI alignment)
        uintptr_t top = ((uintptr_t)(w.stacks + (t + 1) * STACK_BYTES)) & ~(uintptr_t)15;
        void** sp = (void**)(top - 8 * sizeof(void*));
        for (int i = 0; i < 6; ++i) sp[i] = nullptr;
        sp[6] = (void*)&fiber_entry;
        sp[7] = nullptr;
        f.sp = sp;
    }
    w.live = (int)nt;
    w.bar_arrived = 0;
    w.bar_gen = 0;
    unsigned long last_progress = w.progress;
    long idle_passes = 0;
    auto t_idle = std::chrono::steady_clock::now();
    const int sched = g_sched_mode.load(std::memory_order_relaxed);
    std::vector<size_t> order(nw);
    for (size_t i = 0; i < nw; ++i) order[i] = sched == 1 ? nw - 1 - i : i;
    uint64_t rng = 0x9E3779B97F4A7C15ull * (g_sched_seed.load() + 1) + 0xD1B54A32D192ED03ull * (bid.x + 131u * bid.y + 17161u * bid.z + 1);
    while (w.live > 0) {
        if (sched == 2)
            for (size_t i = nw; i > 1; --i) {  // Fisher-Yates with xorshift64*
                rng ^= rng >> 12;
                rng ^= rng << 25;
                rng ^= rng >> 27;
                std::swap(order[i - 1], order[(size_t)((rng * 0x2545F4914F6CDD1Dull) >> 33) % i]);
            }
        for (size_t oi = 0; oi < nw; ++oi)
            for (size_t t = order[oi] * 64, te = std::min(nt, t + 64); t < te; ++t) {
                Fiber& f = w.fibers[t];
                if (f.done) continue;
                w.running = &f;
                cur = &f.ctx;
                hipemu_switch(&w.sched_sp, f.sp);
            }
        if (w.progress != last_progress) {
            last_progress = w.progress;
            idle_passes = 0;
            t_idle = std::chrono::steady_clock::now();
        } else if (++idle_passes > 64) {
            // no barrier released, no collective completed, no thread finished: a deadlock unless the workgroup waits for
            // another workgroup (co-resident mode) - there, give it wall-clock time before giving up
            const double idle_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_idle).count();
            if (!g_cores_now.load() || idle_s > 120.0 || g_abort.load()) {
                char msg[256];
                snprintf(msg, sizeof msg, "hipemu: deadlock in workgroup (%u,%u,%u): %d live threads, %d at the barrier", bid.x,
                         bid.y, bid.z, w.live, w.bar_arrived);
                note_error(msg);
                g_abort.store(true);
                // one more pass lets every waiting fiber see the abort flag and retire
                for (size_t t = 0; t < nt; ++t) {
                    Fiber& f = w.fibers[t];
                    if (f.done) continue;
                    w.running = &f;
                    cur = &f.ctx;
                    hipemu_switch(&w.sched_sp, f.sp);
                }
                return;  // fibers stuck outside a synchronisation point are abandoned
            }
            if (g_cores_now.load()) std::this_thread::yield();
        }
    }
    cur = nullptr;
}

static std::mutex g_count_mu;
static std::atomic<int> g_count_grids{0};   // launch counters keyed by (kernel, workgroups, threads): the under-filled-launch inventory
static std::vector<std::string> g_cores_kernels;  // kernels with grid-wide waits: every workgroup on its own OS thread
static std::vector<std::pair<std::string, long>> g_counts;  // launches per kernel expression (tests assert which kernels ran)

int launch_impl(const char* kernel, U3 grid, U3 block, size_t lds, const std::function<void()>& body) {
    {
        std::lock_guard<std::mutex> lk(g_count_mu);
        bool found = false;
        std::string key_s(kernel);
        if (g_count_grids.load()) key_s += " wg=" + std::to_string((size_t)grid.x * grid.y * grid.z) + " x" + std::to_string(block.x * block.y * block.z);
        for (auto& kv : g_counts)
            if (kv.first == key_s) {
                kv.second++;
                found = true;
                break;
            }
        if (!found) g_counts.emplace_back(key_s, 1);
    }
    const size_t nb = (size_t)grid.x * grid.y * grid.z;
    const size_t nt = (size_t)block.x * block.y * block.z;
    if (nb == 0 || nt == 0 || nt > 1024 || lds > 160 * 1024) {
        note_error("hipemu: invalid launch configuration");
        return hipErrorInvalidValue;
    }
    g_abort.store(false);
    int nthreads = g_threads.load();
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads <= 0) nthreads = 4;
    bool cores = g_coresident.load() != 0;
    {
        std::lock_guard<std::mutex> lk(g_count_mu);
        for (auto& k : g_cores_kernels)
            if (strstr(kernel, k.c_str())) cores = true;
    }
    g_cores_now.store(cores);
    if (cores) {
        if (nb > 512) {
            note_error("hipemu: co-resident launch of more than 512 workgroups");
            return hipErrorInvalidValue;
        }
        nthreads = (int)nb;
    }
    if ((size_t)nthreads > nb) nthreads = (int)nb;
    std::atomic<size_t> next{0};
    auto work = [&]() {
        thread_local Worker worker;
        wk = &worker;
        worker.body = &body;
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= nb || g_abort.load()) break;
            const U3 bid{(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y))};
            run_block(worker, grid, block, bid, lds);
        }
        wk = nullptr;
    };
    if (nthreads == 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int i = 0; i < nthreads; ++i) pool.emplace_back(work);
        for (auto& t : pool) t.join();
    }
    return g_error.load();
}

}  // namespace hipemu

// launches, since the last reset, of kernels whose launch expression contains `substr`
extern "C" __attribute__((visibility("default"))) long hipemu_launch_count(const char* substr) {
    std::lock_guard<std::mutex> lk(hipemu::g_count_mu);
    long n = 0;
    for (auto& kv : hipemu::g_counts)
        if (kv.first.find(substr) != std::string::npos) n += kv.second;
    return n;
}
extern "C" __attribute__((visibility("default"))) void hipemu_print_counts() {
    std::lock_guard<std::mutex> lk(hipemu::g_count_mu);
    long total = 0;
    for (auto& kv : hipemu::g_counts) {
        printf("%6ld  %s\n", kv.second, kv.first.c_str());
        total += kv.second;
    }
    printf("%6ld  launches\n", total);
    fflush(stdout);
}
extern "C" __attribute__((visibility("default"))) void hipemu_count_grids(int on) { hipemu::g_count_grids.store(on); }
extern "C" __attribute__((visibility("default"))) void hipemu_reset_counts() {
    std::lock_guard<std::mutex> lk(hipemu::g_count_mu);
    hipemu::g_counts.clear();
}
extern "C" __attribute__((visibility("default"))) void hipemu_add_coresident_kernel(const char* substr) {
    std::lock_guard<std::mutex> lk(hipemu::g_count_mu);
    hipemu::g_cores_kernels.emplace_back(substr);
}
extern "C" __attribute__((visibility("default"))) void hipemu_set_coresident(int on) { hipemu::g_coresident.store(on); }
extern "C" __attribute__((visibility("default"))) void hipemu_set_wave_schedule(int mode, unsigned seed) {
    hipemu::g_sched_mode.store(mode);
    hipemu::g_sched_seed.store(seed);
}
extern "C" __attribute__((visibility("default"))) void hipemu_set_threads(int n) { hipemu::g_threads.store(n); }
extern "C" __attribute__((visibility("default"))) const char* hipemu_last_message() {
    std::lock_guard<std::mutex> lk(hipemu::g_msg_mu);
    static std::string copy;
    copy = hipemu::g_msg;
    return copy.c_str();
}
