// brotlig_wave_ops.h -- cross-lane primitives for "two 32-lane pages per wave64" (gfx950).
//
// A Brotli-G page is 32 interleaved sub-bitstreams, so its entropy decode is 32 lanes wide
// (the reference shader runs one wave32 per page: src/decoder/BrotliGCompute.hlsl:24-25,:1753).
// A CDNA4 wave is 64 lanes; each wave therefore hosts two independent pages, one per 32-lane
// half.  Every primitive below is either half-scoped (result depends only on the caller's
// 32-lane half) or explicitly wave-scoped.  All of them must be called from wave-uniform
// control flow (all 64 lanes reach the call); per-lane conditions are passed as operands.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Read-only data written by an EARLIER kernel, read through the constant address space: a load with a wave-uniform address is then a scalar
// load (s_load_dword into scalar registers) whatever the compiler can or cannot prove about the kernel's own stores.
#define BROTLIG_CONSTANT_AS __attribute__((address_space(4)))

namespace wave {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
// the same from the execution mask (two instructions), for rare paths that should not keep a register alive for it
__device__ __forceinline__ uint32_t lane_id_fresh()
{
    uint32_t lane;      // (volatile: computed where it stands, never hoisted and carried)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return lane;
}

// ---- wave scope -------------------------------------------------------------------------
// (the i1 form of the ballot: HIP's __ballot(int) widens the predicate to 0 / 1 and compares it again -- a v_cndmask and a
// v_cmp per call on top of the compare that produced the predicate; round 4, ~40 ballots per round)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// Ballots of ONE comparison, written as the instruction they are (v_cmp into a scalar register pair; inactive lanes give 0).  The
// compiler lowers the ballot of a compound predicate -- also one it has folded together itself, e.g. the ballot of n > 32 where n is a
// select -- through a 0 / 1 register and a second compare: two vector-ALU instructions per question, on a kernel bound by the vector
// ALU.  Compound questions are asked as scalar arithmetic on these masks instead (s_and / s_or / s_andn2).
// Round 5: written with the compiler's own intrinsic (llvm.amdgcn.icmp: one v_cmp into a scalar register pair, the same
// instruction as the inline assembly of round 4) -- the intrinsic is modelled as convergent and EXEC-dependent, an `asm` without
// `volatile` was neither: nothing but the call sites' discipline kept the compiler from moving one across a change of the execution mask.
// (predicates: LLVM's ICmpInst numbering -- 32 eq, 33 ne, 34 ugt, 36 ult)
__device__ __forceinline__ uint64_t ballot_gt(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 34); }
__device__ __forceinline__ uint64_t ballot_lt(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 36); }
__device__ __forceinline__ uint64_t ballot_ne(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 33); }
__device__ __forceinline__ uint64_t ballot_eq(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 32); }
template <uint32_t K> __device__ __forceinline__ uint64_t ballot_gt_k(uint32_t a) { return __builtin_amdgcn_uicmp(a, K, 34); }
template <uint32_t K> __device__ __forceinline__ uint64_t ballot_lt_k(uint32_t a) { return __builtin_amdgcn_uicmp(a, K, 36); }
template <uint32_t K> __device__ __forceinline__ uint64_t ballot_eq_k(uint32_t a) { return __builtin_amdgcn_uicmp(a, K, 32); }
__device__ __forceinline__ uint64_t ballot_ne0(uint32_t a) { return __builtin_amdgcn_uicmp(a, 0u, 33); }
__device__ __forceinline__ uint64_t ballot_eq0(uint32_t a) { return __builtin_amdgcn_uicmp(a, 0u, 32); }
// A lane mask (the same in every lane: a ballot, or scalar arithmetic on ballots) as a lane predicate: no instruction, the mask IS the
// condition register.  Lets the compound questions of a loop be asked as s_and / s_andn2 on masks instead of per-lane logic.
__device__ __forceinline__ bool from_mask(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// Value of `v` in lane `src` (0..63) of the wave.
__device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src & 63u) << 2), (int)v);
}

// Shader clock (s_memtime), for the phase-timer diagnostics build.
__device__ __forceinline__ unsigned long long clock() { return (unsigned long long)__builtin_readcyclecounter(); }
// Constant-rate counter (s_memrealtime, 100 MHz), the same on every compute unit: when a wavefront came and went (diagnostics twin).
__device__ __forceinline__ unsigned long long realtime() { return (unsigned long long)__builtin_amdgcn_s_memrealtime(); }

// ---- half scope -------------------------------------------------------------------------
// 32-bit ballot of the caller's half.
__device__ __forceinline__ uint32_t half_ballot(bool p)
{
    const uint64_t m = __builtin_amdgcn_ballot_w64(p);
    return (uint32_t)(m >> (lane_id() & 32u));
}

// The caller's half of a wave-wide mask.
__device__ __forceinline__ uint32_t half_of(uint64_t m) { return (uint32_t)(m >> (lane_id() & 32u)); }

// Value of `v` in lane `src` (0..31) of the caller's half.
__device__ __forceinline__ uint32_t half_shfl(uint32_t v, uint32_t src)
{
    const uint32_t lane = (lane_id() & 32u) | (src & 31u);
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lane << 2), (int)v);
}

// Same, for a source lane that is uniform within each half (the usual case: an index derived
// from a half ballot, or a constant).  Two v_readlane per half instead of a trip through the LDS
// crossbar: no lgkmcnt latency on the critical path.
__device__ __forceinline__ uint32_t half_bcast(uint32_t v, uint32_t src)
{
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)src) & 31u;
    const uint32_t s1 = ((uint32_t)__builtin_amdgcn_readlane((int)src, 32) & 31u) | 32u;
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)s0);
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)s1);
    return lane_id() < 32u ? a : b;
}

// Hand-over between the wavefronts of one workgroup through LDS (one CU, one LDS: accesses are performed in
// the order the LDS unit receives them).  A producer writes its data, then publishes a counter with
// lds_store_release; a consumer polls the counter with lds_load_acquire and then reads the data.
__device__ __forceinline__ uint32_t lds_load_acquire(const uint32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store_release(uint32_t* p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Hand-over between WORKGROUPS of one kernel through global memory (the schedule kernel's tickets, phase counters and the few words its
// phases pass on, brotlig_schedule.h).  The device's XCDs have an L2 each, and ordinary loads and stores are cached there without
// coherence: words that cross workgroups are therefore written and read with AGENT-scope relaxed atomics -- write-through stores, loads
// that look beyond the own L2 -- and ordered by waiting for the own stores (stores_done) before the counter that announces them goes up.
// Round 6's first version used agent-scope release / acquire FENCES instead: each is a write-back or an invalidation of the whole L2
// (buffer_wbl2 / buffer_inv sc1), and two thousand workgroups doing that took 150 us for the work of 20.
__device__ __forceinline__ uint32_t agent_load_relaxed(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t agent_load_relaxed64(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_store_relaxed(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_store_relaxed64(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t agent_add_relaxed(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// compare-and-swap: true when *p was `expect` and is `want` now; otherwise `expect` holds what *p was
__device__ __forceinline__ bool agent_cas64(uint64_t* p, uint64_t& expect, uint64_t want)
{
    return __hip_atomic_compare_exchange_strong(p, &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every memory access this wavefront has issued so far is complete (its write-through stores have arrived) before anything it does afterwards;
// also keeps the compiler from moving memory accesses across it
// the same for code that only one lane runs (no wave-level barrier in it)
__device__ __forceinline__ void lane_stores_done() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
__device__ __forceinline__ void stores_done()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// Give the SIMD to the other wavefronts for a moment (inside a polling loop).
__device__ __forceinline__ void nap() { __builtin_amdgcn_s_sleep(2); }
__device__ __forceinline__ void long_nap() { __builtin_amdgcn_s_sleep(8); }      // ~500 clocks: between looks at a word in global memory

// Issue priority of this wavefront among the wavefronts of its SIMD (0 = default, anything else = raised): raised around a
// round's longest dependent chain, so that the wave that is deepest in latency is served first.
__device__ __forceinline__ void set_priority(int p) { if (p) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }

// Zero, in a scalar register, that the compiler cannot see through.  Used to turn a register-to-register
// copy into an ALU operation (x >> opaque_zero()) where a copy would be placed badly -- see BitReader::refill.
__device__ __forceinline__ uint32_t opaque_zero()
{
    uint32_t z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
}

// A value the caller knows to be the same in every lane, moved to a scalar register.
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The other half's copy of a half-uniform value (two v_readlane).
__device__ __forceinline__ uint32_t other_half(uint32_t v)
{
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
    return lane_id() < 32u ? b : a;
}

// Inclusive prefix sum over the caller's half.  DPP row shifts cover the 16-lane rows;
// row_bcast:15 restricted to rows 1 and 3 carries each even row's total into the odd row
// above it, which closes a 32-lane scan without touching the other half.
__device__ __forceinline__ uint32_t half_scan_incl(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1,3
    return (uint32_t)x;
}

// Same scan through ds_bpermute only (no DPP); used by the device self-test to validate the
// DPP formulation on hardware.
__device__ __forceinline__ uint32_t half_scan_incl_ref(uint32_t v)
{
    const uint32_t sl = lane_id() & 31u;
    uint32_t x = v;
    for (uint32_t d = 1; d < 32; d <<= 1) {
        const uint32_t t = half_shfl(x, sl - d);
        if (sl >= d) x += t;
    }
    return x;
}

__device__ __forceinline__ uint32_t half_sum(uint32_t v) { return half_bcast(half_scan_incl(v), 31); }

// Maximum over the caller's half.
__device__ __forceinline__ uint32_t half_max(uint32_t v)
{
    uint32_t x = v;
    for (uint32_t d = 16; d >= 1; d >>= 1) {
        const uint32_t t = half_shfl(x, (lane_id() & 31u) ^ d);
        x = t > x ? t : x;
    }
    return x;
}

// LDS hand-off between lanes of the wave.  The workgroup is exactly one wave and the LDS
// pipeline executes a wave's DS instructions in issue order, so all this has to do is stop the
// compiler from moving LDS accesses across it.  (__syncthreads() would also work but carries
// s_waitcnt vmcnt(0) + s_barrier, i.e. a full global-memory drain on every call.)
__device__ __forceinline__ void sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Global-memory hand-off between lanes of the wave: earlier stores by any lane of this wave are
// visible to later loads by any lane (one CU, one vector L1; the release drains vmcnt so the
// stores have been performed before the loads issue).
__device__ __forceinline__ void global_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

}  // namespace wave
