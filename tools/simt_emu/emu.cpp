// TEST INFRASTRUCTURE ONLY — fibre runtime of the SIMT emulator (see hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <time.h>

namespace emu {
Fiber* g_cur = nullptr;
uint3_ g_bid{0, 0, 0}, g_bdim{1, 1, 1}, g_gdim{1, 1, 1};
int g_nlive_block = 0;
unsigned long long g_xch[1024][2];

static std::vector<Fiber> g_fibers;
static void* g_main_sp = nullptr;
static const std::function<void()>* g_body = nullptr;
static Rv g_block_rv;
static long g_spin_limit = getenv("EMU_SPIN_LIMIT") ? atol(getenv("EMU_SPIN_LIMIT")) : 20000000L;
static int g_wave_livecnt[16];
static constexpr size_t STACK = 512 << 10;

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");

static void fiber_exit_to_next();
static void wave_try_release(int w);
static void fiber_entry() {
    (*g_body)();
    Fiber* me = g_cur;
    me->done = true;
    g_nlive_block--;
    g_wave_livecnt[me->tid.x >> 6]--;
    wave_try_release((int)(me->tid.x >> 6));
    fiber_exit_to_next();
    abort();
}
// EMU_SCHED=1 runs the lanes in descending order: a kernel whose result changes with the schedule relies on lockstep
// execution somewhere (a cross-lane read or write through LDS without a fence between them)
static int g_dir = (getenv("EMU_SCHED") && atoi(getenv("EMU_SCHED")) == 1) ? -1 : 1;
static int next_live(int from) {
    const int n = (int)g_fibers.size();
    for (int k = 1; k <= n; ++k) {
        const int i = (((from + g_dir * k) % n) + n) % n;
        if (!g_fibers[i].done) return i;
    }
    return -1;
}
static void fiber_exit_to_next() {
    Fiber* me = g_cur;
    const int nx = next_live((int)me->tid.x);
    void* dummy;
    if (nx < 0) { g_cur = nullptr; emu_switch(&dummy, g_main_sp); }
    g_cur = &g_fibers[nx];
    emu_switch(&dummy, g_cur->sp);
}
void yield() {
    Fiber* me = g_cur;
    const int nx = next_live((int)me->tid.x);
    if (nx < 0 || &g_fibers[nx] == me) return;
    g_cur = &g_fibers[nx];
    emu_switch(&me->sp, g_cur->sp);
}
int wave_live(int wave) { return g_wave_livecnt[wave]; }
// Wave-level rendezvous with SIMT divergence: lanes may wait at different source lines (some inside a branch, the
// others already behind it).  When every live lane of the wave waits somewhere, the group at the SMALLEST line is
// released — the lanes inside the branch run first and the others stay parked where control flow joins again, which
// is how the hardware serialises a divergent branch (forward control flow; the kernels' loops are wave-uniform).
struct WaveRv { int wline[64]; int nwait; unsigned long long group; };
static WaveRv g_wrv[16];
static void wave_try_release(int w) {
    WaveRv& r = g_wrv[w];
    if (r.nwait == 0 || r.nwait < g_wave_livecnt[w]) return;
    int m = 0x7fffffff;
    for (int l = 0; l < 64; ++l) if (r.wline[l] >= 0 && r.wline[l] < m) m = r.wline[l];
    unsigned long long g = 0;
    for (int l = 0; l < 64; ++l) if (r.wline[l] == m) { r.wline[l] = -1; r.nwait--; g |= 1ull << l; }
    r.group = g;
}
unsigned long long wave_group() { return g_wrv[g_cur->tid.x >> 6].group; }
void dump_wave_lines() {   // where the lanes of the current wave wait (-1: released / running)
    const WaveRv& r = g_wrv[g_cur->tid.x >> 6];
    fprintf(stderr, "[simt_emu] released group %016llx; lanes waiting at source lines:", r.group);
    for (int l = 0; l < 64; ++l) fprintf(stderr, " %d", r.wline[l]);
    fprintf(stderr, "\n[simt_emu] live count %d, waiting %d; finished lanes:", g_wave_livecnt[g_cur->tid.x >> 6], r.nwait);
    for (int l = 0; l < 64; ++l) if (g_fibers[(g_cur->tid.x & ~63u) + l].done) fprintf(stderr, " %d", l);
    fprintf(stderr, "\n");
}
void rendezvous_wave(int line) {
    const int w = (int)(g_cur->tid.x >> 6), l = (int)(g_cur->tid.x & 63);
    WaveRv& r = g_wrv[w];
    r.wline[l] = line; r.nwait++;
    wave_try_release(w);
    long spins = 0;
    while (r.wline[l] >= 0) {
        yield();
        if (++spins > g_spin_limit) { fprintf(stderr, "[simt_emu] stuck in a wave-level operation at line %d (thread %u)\n", line, g_cur->tid.x); abort(); }
    }
}
static void rendezvous(Rv& rv, int need, int line, const char* what) {
    if (rv.arrived == 0) rv.line = line;
    else if (rv.line != line) {
        fprintf(stderr, "[simt_emu] divergent %s: thread %u at source line %d, others at line %d\n", what, g_cur->tid.x, line, rv.line);
        abort();
    }
    const int gen = rv.gen;
    if (++rv.arrived >= need) { rv.arrived = 0; rv.gen++; return; }
    long spins = 0;
    while (rv.gen == gen) {
        yield();
        if (++spins > 20000000L) { fprintf(stderr, "[simt_emu] deadlock in %s at line %d (thread %u)\n", what, line, g_cur->tid.x); abort(); }
    }
}
void rendezvous_block(int line) { rendezvous(g_block_rv, g_nlive_block, line, "__syncthreads"); }
long long clock_ticks() {
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 100000000LL + ts.tv_nsec / 10;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int nthr = (int)block.x;
    if (nthr > 1024 || block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) { fprintf(stderr, "[simt_emu] unsupported launch shape\n"); abort(); }
    g_bdim = uint3_{block.x, 1, 1};
    g_gdim = uint3_{grid.x, 1, 1};
    g_body = &body;
    if ((int)g_fibers.size() < nthr) {
        const size_t old = g_fibers.size();
        g_fibers.resize(nthr);
        for (size_t i = old; i < g_fibers.size(); ++i) {
            g_fibers[i].stack = (char*)mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (g_fibers[i].stack == (char*)MAP_FAILED) { perror("mmap"); abort(); }
        }
    }
    std::vector<Fiber> saved;   // fibres beyond this launch's thread count stay allocated but idle
    for (unsigned b = 0; b < grid.x; ++b) {
        g_bid = uint3_{b, 0, 0};
        for (int w = 0; w < 16; ++w) { g_wave_livecnt[w] = 0; g_wrv[w].nwait = 0; g_wrv[w].group = 0; for (int l = 0; l < 64; ++l) g_wrv[w].wline[l] = -1; }
        g_block_rv = Rv();
        g_nlive_block = nthr;
        for (int i = 0; i < (int)g_fibers.size(); ++i) {
            Fiber& f = g_fibers[i];
            f.done = i >= nthr;
            if (f.done) continue;
            f.tid = uint3_{(unsigned)i, 0, 0};
            g_wave_livecnt[i >> 6]++;
            // initial frame: six callee-saved registers, then the entry address; rsp is 16-byte aligned at the call boundary
            uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
            void** sp = (void**)(top - 8);          // as if a call had pushed a return address
            *--sp = (void*)&fiber_entry;              // `ret` target
            for (int k = 0; k < 6; ++k) *--sp = nullptr;
            f.sp = sp;
        }
        g_cur = &g_fibers[g_dir > 0 ? 0 : nthr - 1];
        emu_switch(&g_main_sp, g_cur->sp);
    }
    g_body = nullptr;
}
}  // namespace emu
