// sim_runtime.h -- TEST INFRASTRUCTURE: a tiny CPU stand-in for the HIP execution model so the
// kernel source in brotli_g_sdk_amd/csrc/brotlig_kernels.h can be executed and debugged on a box
// without a GPU.  One workgroup (one or more wave64s) runs at a time; each lane is a fiber; every
// cross-lane primitive in tests/sim/brotlig_wave_ops.h is a rendezvous of all live lanes of the wave that also
// checks that every lane arrived from the same call site (the kernel's wave-uniformity rule);
// __syncthreads() is a rendezvous of all live lanes of all waves of the workgroup.
// Nothing here is part of the product, and the product never includes this file.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

namespace sim {

constexpr int kLanes = 64;
struct Dim3 { uint32_t x, y, z; };

struct Fiber {
    void* sp;
    void* stack;
    bool  done;
    // rendezvous state
    uint64_t gen_seen;
};

struct WaveState {
    Fiber    fiber[kLanes];
    void*    sched_sp;
    int      cur;                 // running lane
    uint32_t block;               // blockIdx.x
    uint32_t grid;
    void   (*body)(void*);
    void*    arg;
    // collective
    uint64_t gen;                 // completed collectives
    uint64_t in_a[2][kLanes], in_b[2][kLanes];
    int      site[kLanes];
    bool     waiting[kLanes];
    bool     at_barrier[kLanes];  // parked in __syncthreads()
    uint64_t n_collectives;
    int      index;               // wave number inside the workgroup
};

constexpr int kMaxWaves = 8;
extern WaveState g_waves[kMaxWaves];
extern WaveState* g_cw;           // the wave whose fiber is running
extern uint64_t g_barrier_gen;    // completed workgroup barriers
inline WaveState& cw() { return *g_cw; }
#define g_wave cw()

extern "C" void sim_switch(void** from_sp, void* to_sp);

inline void yield_to_scheduler() { WaveState& w = cw(); sim_switch(&w.fiber[w.cur].sp, w.sched_sp); }

// Deposit operands, wait until every live lane has done so, return the parity slot to read.
inline int collective_enter(uint64_t a, uint64_t b, int site)
{
    WaveState& w = g_wave;
    const int lane = w.cur;
    const int slot = (int)(w.gen & 1);
    const uint64_t my_gen = w.gen;
    w.in_a[slot][lane] = a; w.in_b[slot][lane] = b;
    w.site[lane] = site; w.waiting[lane] = true;
    while (w.gen == my_gen) yield_to_scheduler();
    return slot;
}

// Park the running fiber until every live lane of every wave of the workgroup has arrived.
inline void barrier_enter(int site)
{
    WaveState& w = cw();
    const int lane = w.cur;
    const uint64_t my_gen = g_barrier_gen;
    w.site[lane] = site; w.at_barrier[lane] = true;
    while (g_barrier_gen == my_gen) yield_to_scheduler();
}

void run_grid(uint32_t grid, void (*body)(void*), void* arg, int waves = 1);

inline uint32_t lane_now() { return (uint32_t)cw().cur; }
inline uint32_t thread_now() { return (uint32_t)(cw().index * kLanes + cw().cur); }

}  // namespace sim

// ---- HIP built-ins used by the kernels --------------------------------------------------------
struct SimThreadIdx { uint32_t y = 0, z = 0; struct X { operator uint32_t() const { return sim::thread_now(); } } x; };
namespace sim { extern uint32_t g_block_y, g_grid_y; }     // the y dimension of the launch being simulated (set by the harness around run_grid; 0 of 1 by default)
struct SimBlockIdx { uint32_t z = 0; struct X { operator uint32_t() const { return sim::cw().block; } } x; struct Y { operator uint32_t() const { return sim::g_block_y; } } y; };
struct SimGridDim { uint32_t z = 1; struct X { operator uint32_t() const { return sim::cw().grid; } } x; struct Y { operator uint32_t() const { return sim::g_grid_y; } } y; };
struct SimBlockDim { uint32_t y = 1, z = 1; uint32_t x = sim::kLanes; };
static const SimBlockDim blockDim;
static const SimThreadIdx threadIdx;
static const SimBlockIdx blockIdx;
static const SimGridDim gridDim;

inline int __popc(uint32_t x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline uint32_t __brev(uint32_t x)
{
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
}
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
inline uint32_t atomicOr(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
