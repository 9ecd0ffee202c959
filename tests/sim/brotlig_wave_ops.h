// TEST INFRASTRUCTURE: CPU-simulator implementation of the wave:: primitives declared by
// brotli_g_sdk_amd/csrc/brotlig_wave_ops.h (same names, same semantics).  Selected by putting
// tests/sim in front of the include path when building tests/sim/sim_decode.cpp.
#pragma once
#include "sim_runtime.h"

#define BROTLIG_CONSTANT_AS

namespace wave {

inline uint32_t lane_id() { return sim::lane_now() & 63u; }
inline uint32_t lane_id_fresh() { return lane_id(); }

#define SIM_SITE int site = __builtin_LINE()

inline unsigned long long clock() { return (unsigned long long)__builtin_ia32_rdtsc(); }
inline unsigned long long realtime() { return (unsigned long long)__builtin_ia32_rdtsc(); }
inline uint64_t ballot64(bool p, SIM_SITE)
{
    const int s = sim::collective_enter(p ? 1 : 0, 0, site);
    uint64_t m = 0;
    for (int l = 0; l < sim::kLanes; ++l) if (!sim::cw().fiber[l].done && sim::cw().in_a[s][l]) m |= 1ull << l;
    return m;
}
inline bool any(bool p, SIM_SITE) { return ballot64(p, site) != 0; }
inline uint64_t ballot_gt(uint32_t a, uint32_t b, SIM_SITE) { return ballot64(a > b, site); }
inline uint64_t ballot_lt(uint32_t a, uint32_t b, SIM_SITE) { return ballot64(a < b, site); }
inline uint64_t ballot_ne(uint32_t a, uint32_t b, SIM_SITE) { return ballot64(a != b, site); }
inline uint64_t ballot_eq(uint32_t a, uint32_t b, SIM_SITE) { return ballot64(a == b, site); }
template <uint32_t K> inline uint64_t ballot_gt_k(uint32_t a, SIM_SITE) { return ballot64(a > K, site); }
template <uint32_t K> inline uint64_t ballot_lt_k(uint32_t a, SIM_SITE) { return ballot64(a < K, site); }
template <uint32_t K> inline uint64_t ballot_eq_k(uint32_t a, SIM_SITE) { return ballot64(a == K, site); }
inline uint64_t ballot_ne0(uint32_t a, SIM_SITE) { return ballot64(a != 0u, site); }
inline uint64_t ballot_eq0(uint32_t a, SIM_SITE) { return ballot64(a == 0u, site); }
inline bool from_mask(uint64_t m) { return ((m >> lane_id()) & 1u) != 0u; }
inline uint32_t half_of(uint64_t m) { return (uint32_t)(m >> (lane_id() & 32u)); }
inline uint32_t bcast(uint32_t v, uint32_t src, SIM_SITE)
{
    const int s = sim::collective_enter(v, 0, site);
    return (uint32_t)sim::cw().in_a[s][src & 63u];
}
inline uint32_t uniform(uint32_t v) { return v; }
inline uint32_t opaque_zero() { return 0u; }
inline void set_priority(int) {}
inline uint32_t lds_load_acquire(const uint32_t* p) { return *(const volatile uint32_t*)p; }
inline void lds_store_release(uint32_t* p, uint32_t v) { *(volatile uint32_t*)p = v; }
inline void nap() { sim::yield_to_scheduler(); }      // let the other wave of the workgroup run
// between workgroups (the schedule kernel): the simulator runs one workgroup at a time, in blockIdx order -- whatever a workgroup waits for
// was finished by an earlier one, or the kernel's ordering argument is broken (the polling loops of brotlig_schedule.h give up and say so)
inline uint32_t agent_load_relaxed(const uint32_t* p) { return *(const volatile uint32_t*)p; }
inline uint64_t agent_load_relaxed64(const uint64_t* p) { return *(const volatile uint64_t*)p; }
inline void agent_store_relaxed(uint32_t* p, uint32_t v) { *(volatile uint32_t*)p = v; }
inline void agent_store_relaxed64(uint64_t* p, uint64_t v) { *(volatile uint64_t*)p = v; }
inline uint32_t agent_add_relaxed(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
inline bool agent_cas64(uint64_t* p, uint64_t& expect, uint64_t want) { if (*p == expect) { *p = want; return true; } expect = *p; return false; }
inline void long_nap() {}
inline void lane_stores_done() {}
inline void stores_done(int site = __builtin_LINE()) { sim::collective_enter(0, 0, site); }
inline uint32_t other_half(uint32_t v, SIM_SITE)
{
    const int s = sim::collective_enter(v, 0, site);
    return (uint32_t)sim::cw().in_a[s][lane_id() < 32u ? 32 : 0];
}
inline uint32_t half_ballot(bool p, SIM_SITE) { return (uint32_t)(ballot64(p, site) >> (lane_id() & 32u)); }
inline uint32_t half_shfl(uint32_t v, uint32_t src, SIM_SITE)
{
    const int s = sim::collective_enter(v, 0, site);
    return (uint32_t)sim::cw().in_a[s][(lane_id() & 32u) | (src & 31u)];
}
inline uint32_t half_bcast(uint32_t v, uint32_t src, SIM_SITE)
{
    // the device version requires `src` to be uniform within each half: check it
    const int s = sim::collective_enter(v, src, site);
    const uint32_t base = lane_id() & 32u;
    for (uint32_t l = base; l < base + 32u; ++l)
        if (!sim::cw().fiber[l].done && sim::cw().in_b[s][l] != src) {
            fprintf(stderr, "SIM: half_bcast with a non-uniform source lane at line %d\n", site); abort();
        }
    return (uint32_t)sim::cw().in_a[s][base | (src & 31u)];
}
inline uint32_t half_scan_incl(uint32_t v, SIM_SITE)
{
    const int s = sim::collective_enter(v, 0, site);
    const uint32_t me = lane_id(), base = me & 32u;
    uint32_t acc = 0;
    for (uint32_t l = base; l <= me; ++l) acc += (uint32_t)sim::cw().in_a[s][l];
    return acc;
}
inline uint32_t half_scan_incl_ref(uint32_t v, SIM_SITE) { return half_scan_incl(v, site); }
inline uint32_t half_sum(uint32_t v, SIM_SITE)
{
    const int s = sim::collective_enter(v, 0, site);
    const uint32_t base = lane_id() & 32u;
    uint32_t acc = 0;
    for (uint32_t l = base; l < base + 32u; ++l) acc += (uint32_t)sim::cw().in_a[s][l];
    return acc;
}
inline uint32_t half_max(uint32_t v, SIM_SITE)
{
    const int s = sim::collective_enter(v, 0, site);
    const uint32_t base = lane_id() & 32u;
    uint32_t acc = 0;
    for (uint32_t l = base; l < base + 32u; ++l) { const uint32_t x = (uint32_t)sim::cw().in_a[s][l]; acc = x > acc ? x : acc; }
    return acc;
}
inline void sync(SIM_SITE) { sim::collective_enter(0, 0, site); }
inline void global_fence(SIM_SITE) { sim::collective_enter(0, 0, site); }

#undef SIM_SITE
}  // namespace wave

inline void __syncthreads(int site = __builtin_LINE()) { sim::barrier_enter(site); }
