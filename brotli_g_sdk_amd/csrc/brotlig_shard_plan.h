// brotlig_shard_plan.h -- the multi-device shard plan (include/brotlig_amd.h: BrotligShardPlan), pure host arithmetic.
// One definition, exported by libbrotlig_hip.so and by libbrotlig_cpu.so: a host without ROCm (a gloo / CPU-only rank, a
// scheduler process) computes the same cut as the ranks that decode.
//
// Streams are independent, so a batch shards with no exchange between devices: contiguous runs of streams, one run per
// device, cut so that the largest run's COMPRESSED bytes are as small as possible (the cost of a page follows its
// compressed size, not its 64 KiB of output; SURVEY.md 8(e) last row).  Streams are never split (a pre-conditioned stream's
// pages scatter over its whole texture).  The reference's analogue is the fan-out of pages over host threads,
// src/BrotligDecoder.cpp:356-375, and of streams over the shader's queue, BrotliGCompute.hlsl:1757-1881.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace brotlig {

// first[g] .. first[g + 1] - 1 = shard g (first has num_shards + 1 entries).  Returns false on bad arguments.
// 1. the bottleneck: the smallest cap such that a left-to-right packing needs at most num_shards runs (binary search);
// 2. among the cuts that respect it, every run ends where its weight is closest to an even share of what is left (so ten
//    equal streams over four devices come out 3,2,3,2 rather than 3,3,3,1), never leaving a later shard without a stream
//    while streams remain.
inline bool shard_plan(const uint64_t* in_sizes, uint32_t num_streams, uint32_t num_shards, uint32_t* first)
{
    if (!in_sizes || !first || num_shards == 0u) return false;
    const uint32_t n = num_streams;
    std::vector<uint64_t> pre(n + 1u, 0);
    uint64_t lo = 0;
    for (uint32_t i = 0; i < n; ++i) { pre[i + 1u] = pre[i] + in_sizes[i]; lo = std::max(lo, in_sizes[i]); }
    uint64_t hi = pre[n];
    // end of the longest run that starts at i and weighs at most cap (at least one stream)
    auto run_end = [&](uint32_t i, uint64_t cap) {
        const uint32_t e = (uint32_t)(std::upper_bound(pre.begin() + i + 1, pre.end(), pre[i] + cap) - pre.begin()) - 1u;
        return std::max(e, i + 1u);
    };
    auto runs_needed = [&](uint64_t cap) { uint32_t runs = 0; for (uint32_t i = 0; i < n; i = run_end(i, cap)) ++runs; return runs; };
    while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2u; if (runs_needed(mid) <= num_shards) hi = mid; else lo = mid + 1u; }
    const uint64_t cap = lo;
    std::vector<uint32_t> need(n + 1u, 0);          // runs needed for streams i.. under the cap
    for (uint32_t i = n; i-- > 0u;) need[i] = 1u + need[run_end(i, cap)];
    uint32_t i = 0;
    for (uint32_t g = 0; g < num_shards; ++g) {
        first[g] = i;
        const uint32_t after = num_shards - 1u - g;
        if (i >= n) continue;
        if (after == 0u) { i = n; continue; }
        // feasible ends e: weight <= cap, the rest still fits the shards after me, and (while possible) a stream for each of them
        const uint32_t e_max = std::min(run_end(i, cap), n - std::min(after, n - i - 1u));
        uint32_t e_min = i + 1u;
        while (e_min < e_max && need[e_min] > after) ++e_min;
        const uint64_t left = pre[n] - pre[i];
        const uint64_t ideal = (left + (num_shards - g) - 1u) / (num_shards - g);
        uint32_t best = e_min;
        for (uint32_t e = e_min; e <= e_max; ++e) {
            const uint64_t w = pre[e] - pre[i], wb = pre[best] - pre[i];
            const uint64_t d = w > ideal ? w - ideal : ideal - w, db = wb > ideal ? wb - ideal : ideal - wb;
            if (d <= db) best = e;
        }
        i = best;
    }
    first[num_shards] = n;
    return true;
}

}  // namespace brotlig
