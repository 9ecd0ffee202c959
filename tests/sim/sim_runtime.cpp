// TEST INFRASTRUCTURE: fiber scheduler behind sim_runtime.h (x86-64 SysV only).
#include "sim_runtime.h"

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

namespace sim {

WaveState g_wave;

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
    WaveState& w = g_wave;
    w.body(w.arg);
    w.fiber[w.cur].done = true;
    for (;;) yield_to_scheduler();
}

constexpr size_t kStackBytes = 512 * 1024;

// crash diagnostics: which lane, after which cross-lane call site, and a raw backtrace
static void on_segv(int)
{
    char msg[128];
    int n = snprintf(msg, sizeof msg, "SIM: SIGSEGV in lane %d (block %u), last cross-lane site line %d\n",
                     g_wave.cur, g_wave.block, g_wave.site[g_wave.cur]);
    (void)!write(2, msg, (size_t)n);
    void* bt[32];
    backtrace_symbols_fd(bt, backtrace(bt, 32), 2);
    _exit(139);
}

void run_grid(uint32_t grid, void (*body)(void*), void* arg)
{
    WaveState& w = g_wave;
    static bool hooked = false;
    if (!hooked) { hooked = true; static char alt[65536]; stack_t ss{alt, 0, sizeof alt}; sigaltstack(&ss, nullptr);
        struct sigaction sa{}; sa.sa_handler = on_segv; sa.sa_flags = SA_ONSTACK; sigaction(SIGSEGV, &sa, nullptr); }
    w.grid = grid; w.body = body; w.arg = arg;
    static void* stacks[kLanes] = {nullptr};
    for (int l = 0; l < kLanes; ++l) if (!stacks[l]) stacks[l] = aligned_alloc(64, kStackBytes);
    for (uint32_t b = 0; b < grid; ++b) {
        w.block = b; w.gen = 0;
        for (int l = 0; l < kLanes; ++l) {
            Fiber& f = w.fiber[l];
            f.stack = stacks[l]; f.done = false; w.waiting[l] = false;
            uintptr_t top = ((uintptr_t)stacks[l] + kStackBytes) & ~(uintptr_t)15;
            void** sp = (void**)top;
            *--sp = nullptr;                    // fake return address of fiber_entry
            *--sp = (void*)&fiber_entry;        // popped by `ret`
            for (int r = 0; r < 6; ++r) *--sp = nullptr;
            f.sp = sp;
        }
        for (;;) {
            int live = 0, waiting = 0;
            for (int l = 0; l < kLanes; ++l) {
                if (w.fiber[l].done) continue;
                if (!w.waiting[l]) { w.cur = l; sim_switch(&w.sched_sp, w.fiber[l].sp); }
                if (!w.fiber[l].done) { ++live; if (w.waiting[l]) ++waiting; }
            }
            if (live == 0) break;
            if (waiting != live) continue;      // some lane still runnable (cannot happen: lanes run to a wait)
            // every live lane is parked in a collective: they must agree on the call site
            int site = -1;
            for (int l = 0; l < kLanes; ++l) {
                if (w.fiber[l].done) continue;
                if (site < 0) site = w.site[l];
                else if (site != w.site[l]) {
                    fprintf(stderr, "SIM: divergent cross-lane op: lane %d at line %d, earlier lanes at line %d (block %u)\n",
                            l, w.site[l], site, b);
                    abort();
                }
            }
            for (int l = 0; l < kLanes; ++l) w.waiting[l] = false;
            ++w.gen; ++w.n_collectives;
        }
    }
}

}  // namespace sim
