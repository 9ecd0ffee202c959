// TEST INFRASTRUCTURE: fiber scheduler behind sim_runtime.h (x86-64 SysV only).
#include "sim_runtime.h"

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

namespace sim {
uint32_t g_block_y = 0, g_grid_y = 1;

WaveState g_waves[kMaxWaves];
WaveState* g_cw = &g_waves[0];
uint64_t g_barrier_gen = 0;

// Minimal context switch: callee-saved registers + stack pointer.
asm(R"(
.text
.globl sim_switch
.type sim_switch,@function
sim_switch:
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
.size sim_switch, .-sim_switch
)");

static void fiber_entry()
{
    {
        WaveState& w = cw();
        w.body(w.arg);
    }
    cw().fiber[cw().cur].done = true;
    for (;;) yield_to_scheduler();
}

constexpr size_t kStackBytes = 512 * 1024;

// crash diagnostics: which lane, after which cross-lane call site, and a raw backtrace
static void on_segv(int)
{
    char msg[128];
    int n = snprintf(msg, sizeof msg, "SIM: SIGSEGV in wave %d lane %d (block %u), last cross-lane site line %d\n",
                     cw().index, cw().cur, cw().block, cw().site[cw().cur]);
    (void)!write(2, msg, (size_t)n);
    void* bt[32];
    backtrace_symbols_fd(bt, backtrace(bt, 32), 2);
    _exit(139);
}

void run_grid(uint32_t grid, void (*body)(void*), void* arg, int waves)
{
    static bool hooked = false;
    if (!hooked) { hooked = true; static char alt[65536]; stack_t ss{alt, 0, sizeof alt}; sigaltstack(&ss, nullptr);
        struct sigaction sa{}; sa.sa_handler = on_segv; sa.sa_flags = SA_ONSTACK; sigaction(SIGSEGV, &sa, nullptr); }
    if (waves < 1 || waves > kMaxWaves) { fprintf(stderr, "SIM: %d waves per workgroup not supported\n", waves); abort(); }
    static void* stacks[kMaxWaves][kLanes] = {{nullptr}};
    for (int v = 0; v < waves; ++v)
        for (int l = 0; l < kLanes; ++l) if (!stacks[v][l]) stacks[v][l] = aligned_alloc(64, kStackBytes);
    for (uint32_t b = 0; b < grid; ++b) {
        for (int v = 0; v < waves; ++v) {
            WaveState& w = g_waves[v];
            w.grid = grid; w.body = body; w.arg = arg; w.block = b; w.gen = 0; w.index = v;
            for (int l = 0; l < kLanes; ++l) {
                Fiber& f = w.fiber[l];
                f.stack = stacks[v][l]; f.done = false; w.waiting[l] = false; w.at_barrier[l] = false;
                uintptr_t top = ((uintptr_t)stacks[v][l] + kStackBytes) & ~(uintptr_t)15;
                void** sp = (void**)top;
                *--sp = nullptr;                    // fake return address of fiber_entry
                *--sp = (void*)&fiber_entry;        // popped by `ret`
                for (int r = 0; r < 6; ++r) *--sp = nullptr;
                f.sp = sp;
            }
        }
        for (;;) {
            int live_total = 0, at_barrier_total = 0;
            bool progress = false;
            for (int v = 0; v < waves; ++v) {
                WaveState& w = g_waves[v];
                g_cw = &w;
                int live = 0, waiting = 0, barrier = 0;
                for (int l = 0; l < kLanes; ++l) {
                    if (w.fiber[l].done) continue;
                    if (!w.waiting[l] && !w.at_barrier[l]) { w.cur = l; sim_switch(&w.sched_sp, w.fiber[l].sp); progress = true; }
                    if (!w.fiber[l].done) { ++live; if (w.waiting[l]) ++waiting; if (w.at_barrier[l]) ++barrier; }
                }
                live_total += live; at_barrier_total += barrier;
                if (live == 0 || waiting == 0) continue;
                if (waiting != live) {
                    if (waiting + barrier == live) {
                        fprintf(stderr, "SIM: wave %d: %d lanes in a cross-lane op while %d lanes wait at a workgroup barrier (block %u)\n",
                                v, waiting, barrier, b);
                        abort();
                    }
                    continue;
                }
                // every live lane of this wave is parked in a collective: they must agree on the call site
                int site = -1;
                for (int l = 0; l < kLanes; ++l) {
                    if (w.fiber[l].done) continue;
                    if (site < 0) site = w.site[l];
                    else if (site != w.site[l]) {
                        fprintf(stderr, "SIM: divergent cross-lane op: wave %d lane %d at line %d, earlier lanes at line %d (block %u)\n",
                                v, l, w.site[l], site, b);
                        abort();
                    }
                }
                for (int l = 0; l < kLanes; ++l) w.waiting[l] = false;
                ++w.gen; ++w.n_collectives;
                progress = true;
            }
            if (live_total == 0) break;
            if (at_barrier_total == live_total) {                       // the whole workgroup has arrived
                for (int v = 0; v < waves; ++v) for (int l = 0; l < kLanes; ++l) g_waves[v].at_barrier[l] = false;
                ++g_barrier_gen;
                progress = true;
            }
            if (!progress) {
                fprintf(stderr, "SIM: deadlock in block %u: %d live lanes, %d at the workgroup barrier, the rest gone or stuck\n",
                        b, live_total, at_barrier_total);
                abort();
            }
        }
    }
    g_cw = &g_waves[0];
}

}  // namespace sim
