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

namespace wave {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// ---- wave scope -------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }
__device__ __forceinline__ bool any(bool p) { return __ballot(p) != 0ull; }
// Value of `v` in lane `src` (0..63) of the wave.
__device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src & 63u) << 2), (int)v);
}

// ---- half scope -------------------------------------------------------------------------
// 32-bit ballot of the caller's half.
__device__ __forceinline__ uint32_t half_ballot(bool p)
{
    const uint64_t m = __ballot(p);
    return (uint32_t)(m >> (lane_id() & 32u));
}

// Value of `v` in lane `src` (0..31) of the caller's half.
__device__ __forceinline__ uint32_t half_shfl(uint32_t v, uint32_t src)
{
    const uint32_t lane = (lane_id() & 32u) | (src & 31u);
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lane << 2), (int)v);
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

__device__ __forceinline__ uint32_t half_sum(uint32_t v) { return half_shfl(half_scan_incl(v), 31); }

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

// LDS hand-off between lanes of the wave (workgroup == one wave).
__device__ __forceinline__ void sync() { __syncthreads(); }

// Global-memory hand-off between lanes of the wave: earlier stores by any lane of this
// workgroup are visible to later loads by any lane (same CU, shared vector L1).
__device__ __forceinline__ void global_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

}  // namespace wave
