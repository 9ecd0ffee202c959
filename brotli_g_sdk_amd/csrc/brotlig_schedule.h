// brotlig_schedule.h -- the kernels in front of the page decode: stream headers -> page counts -> prefix, the page schedule, the pairing policy.
// Part of the gfx950 Brotli-G decode kernels; brotlig_kernels.h includes the parts in order and says what the whole replaces.
#pragma once
#include "brotlig_jobs.h"
#include "brotlig_decondition.h"

namespace brotlig {

// -------------------------------------------------------------------------------------------
// Kernel 1: page counts per stream -> exclusive prefix.  One workgroup per 64 streams.  A batch of up to 64 streams is done in this one
// launch; for more, every workgroup leaves the prefix inside its 64 streams and their page total (DcTable::chunk_pages of its first stream),
// and brotlig_prepare_finish_kernel adds what lies before.  (Rounds 1-4 walked all streams in ONE workgroup, 64 per step, every step a chain
// of dependent loads -- descriptor, header, table: 1.3 ms for a batch of 65 536 small streams, a quarter of its whole decode; round 5,
// profiles/experiments/r05_many_streams.md.)
__global__ void __launch_bounds__(64) brotlig_prepare_kernel(DecodeArgs a)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t s = blockIdx.x * 64u + lane;
    uint32_t pages = 0, supers = 0;
    if (s < a.num_streams) {
        const uint8_t* p = a.in + a.streams[s].in_offset;
        StreamInfo si;
        const uint64_t in_off = a.streams[s].in_offset;
        const uint64_t in_end = stream_in_end(a.streams[s], a.in_bytes), out_end = stream_out_end(a.streams[s], a.out_bytes);
        const bool hdr_in = in_off + 8u <= in_end;
        // Beyond the reference's two checks (src/BrotligDecoder.cpp:437-446): the page table must lie inside the
        // stream, a short last page cannot be longer than a page, and the stream's pages must fit the region the
        // caller gave it -- a damaged header must not send pages into a neighbouring stream's output.
        bool ok = hdr_in && parse_stream_header(load_u32(p), load_u32(p + 4), si) &&
                  in_off + si.header_bytes + 4ull * si.num_pages <= in_end && si.last_page_size <= si.page_size;
        uint64_t usz = 0;
        if (ok) {
            usz = (uint64_t)si.num_pages * si.page_size - (si.last_page_size ? si.page_size - si.last_page_size : 0u);
            ok = a.streams[s].out_offset + usz <= out_end;
        }
        if (ok) pages = si.num_pages;
        else atomicOr(a.status, kStatusBadHeader);
        DcTable& t = a.dc[s];
        t.precon = 0;
        t.status = ok ? 0u : kStatusBadHeader;                      // the stream's own status word (pages add kStatusBadPage)
        if (pages && si.preconditioned) {
            // the texture described by the precondition header is the stream's output (:478): the de-conditioning
            // kernel writes all of it, whatever happened to the stream's pages
            if (a.scratch == nullptr || usz > 0xFFFFFFFFull || !dc_init(t, load_u32(p + 8), load_u32(p + 12), (uint32_t)usz)) {
                t.precon = 0; pages = 0;                            // the reference has undefined behaviour here
                t.status = kStatusBadHeader;
                atomicOr(a.status, kStatusBadHeader);
            } else { atomicAdd(a.status + 2, 1u); supers = t.item_prefix[t.num_mips] >> 8; }
        }
    }
    const uint32_t lo = wave::half_scan_incl(pages);
    const uint32_t lo_total = wave::half_bcast(lo, 31);
    const uint32_t first_half_total = wave::bcast(lo_total, 0);
    const uint32_t second_half_total = wave::bcast(lo_total, 32);
    const uint32_t incl = lane < 32u ? lo : lo + first_half_total;
    const uint32_t total = first_half_total + second_half_total;
    // the same for the de-conditioning super-tiles
    const uint32_t su = wave::half_scan_incl(supers);
    const uint32_t su_total = wave::half_bcast(su, 31);
    const uint32_t su_first = wave::bcast(su_total, 0), su_second = wave::bcast(su_total, 32);
    const uint32_t su_incl = lane < 32u ? su : su + su_first;
    if (s < a.num_streams) { a.page_base[s] = incl - pages; a.dc[s].super_base = su_incl - supers; }
    if (lane == 0u) {
        if (gridDim.x == 1u) { a.page_base[a.num_streams] = total; a.work_counter[0] = 0u; a.status[5] = su_first + su_second; }
        else { a.dc[s].chunk_pages = total; a.dc[s].chunk_supers = su_first + su_second; }
    }
}

// Kernel 1b (batches of more than 64 streams; same grid): the pages of all earlier workgroups' streams, added to this one's 64 entries.
__global__ void __launch_bounds__(64) brotlig_prepare_finish_kernel(DecodeArgs a)
{
    const uint32_t lane = threadIdx.x, c = blockIdx.x;
    uint32_t acc = 0, acc_su = 0;
    for (uint32_t j = lane; j < c; j += 64u) { acc += a.dc[j * 64u].chunk_pages; acc_su += a.dc[j * 64u].chunk_supers; }
    const uint32_t lo = wave::half_scan_incl(acc);
    const uint32_t lo_total = wave::half_bcast(lo, 31);
    const uint32_t before = wave::bcast(lo_total, 0) + wave::bcast(lo_total, 32);
    const uint32_t su = wave::half_scan_incl(acc_su);
    const uint32_t su_total = wave::half_bcast(su, 31);
    const uint32_t before_su = wave::bcast(su_total, 0) + wave::bcast(su_total, 32);
    const uint32_t s = c * 64u + lane;
    if (s < a.num_streams) { a.page_base[s] += before; a.dc[s].super_base += before_su; }
    if (c + 1u == gridDim.x && lane == 0u) {
        a.page_base[a.num_streams] = before + a.dc[c * 64u].chunk_pages; a.work_counter[0] = 0u;
        a.status[5] = before_su + a.dc[c * 64u].chunk_supers;
    }
}

// -------------------------------------------------------------------------------------------
// Page schedule.  The decode kernel runs two pages per wavefront and pays the maximum of the two in
// every phase of a round, so it matters which pages meet: the same 4 GiB of mixed pages decode 12 %
// faster when similar pages are neighbours.  Pages are therefore grouped into buckets by
// compressed size relative to the page size (an eighth of an octave per bucket since round 5, see below; stored pages last) and
// handed out bucket by bucket, dense pages first (they are the slow ones, which also shortens the
// tail of the launch).  Two passes over the page tables: count, then scatter into `order`.

// compressed and decompressed size of global page g (same walk as fetch_job)
__device__ inline void page_sizes(const DecodeArgs& a, uint32_t g, uint32_t total, uint32_t& in_size, uint32_t& out_size)
{
    uint32_t lo = 0, hi = a.num_streams;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (a.page_base[mid] <= g) lo = mid; else hi = mid; }
    const uint32_t i = g - a.page_base[lo];
    const uint32_t np = (lo + 1u < a.num_streams ? a.page_base[lo + 1u] : total) - a.page_base[lo];
    const uint8_t* sp = a.in + a.streams[lo].in_offset;
    StreamInfo si;
    parse_stream_header(load_u32(sp), load_u32(sp + 4), si);
    const uint8_t* table = sp + si.header_bytes;
    const uint32_t off = i ? load_u32(table + 4u * i) : 0u;
    in_size = i + 1u < np ? load_u32(table + 4u * (i + 1u)) - off : load_u32(table);
    out_size = (i + 1u == np && si.last_page_size) ? si.last_page_size : si.page_size;
}
// Round 5: the buckets are a quarter / an eighth of an octave wide instead of half an octave (BROTLIG_TUNE_BUCKETS_PER_OCTAVE).  On the mixed
// benchmark the pairing policy below keeps the two halves of a wavefront in step (a free half waits for its neighbour), so what a pair costs
// is its SLOWER page: the narrower the bucket, the closer two neighbours of the schedule are in compressed size -- the one cost signal a page
// table holds.  (The data classes of the benchmark already sat in buckets of their own: text 3, samples16 4, records 6-9, runs 12-13 of 16.)
#ifndef BROTLIG_TUNE_BUCKETS_PER_OCTAVE
#define BROTLIG_TUNE_BUCKETS_PER_OCTAVE 8
#endif
constexpr uint32_t kBucketsPerOctave = BROTLIG_TUNE_BUCKETS_PER_OCTAVE;
static_assert(kBucketsPerOctave == 2u || kBucketsPerOctave == 4u || kBucketsPerOctave == 8u, "half, quarter or eighth octaves");
constexpr uint32_t kBuckets = 8u * kBucketsPerOctave;                   // 7.5 .. 7.9 octaves of compression ratio, then "denser", then "stored"
constexpr uint32_t kBucketStep16 = kBucketsPerOctave == 2u ? 46341u : kBucketsPerOctave == 4u ? 55109u : 60097u;    // 2^(-1/n) in 16-bit fixed point
__device__ __forceinline__ uint32_t page_bucket(uint32_t in_size, uint32_t out_size)
{
    if (in_size >= out_size) return kBuckets - 1u;                      // stored (or nonsense): cheapest, last
    // bucket b holds in_size in (out / 2^((b+1)/n), out / 2^(b/n)]
    uint32_t b = 0, t = (uint32_t)(((uint64_t)out_size * kBucketStep16) >> 16);     // out_size <= 128 KiB
    while (b < kBuckets - 2u && in_size <= t) { ++b; t = (uint32_t)(((uint64_t)t * kBucketStep16) >> 16); }
    return b;
}
constexpr uint32_t kOrderHist = 8, kOrderCursor = 8 + kBuckets;         // status word offsets
constexpr uint32_t kStatusWords = 8u + 2u * 64u;                        // the workspace header: room for the widest setting
static_assert(kOrderCursor + kBuckets <= kStatusWords && kBuckets <= 64u, "status words; one lane per bucket in the order kernels");

__global__ void __launch_bounds__(64) brotlig_order_count_kernel(DecodeArgs a)
{
    __shared__ uint32_t hist[kBuckets];
    const uint32_t lane = threadIdx.x, total = a.page_base[a.num_streams];
    if (a.order == nullptr || total > a.order_cap) return;
    if (lane < kBuckets) hist[lane] = 0u;
    wave::sync();
    for (uint32_t g = blockIdx.x * 64u + lane; g < total; g += gridDim.x * 64u) {
        uint32_t in_size, out_size;
        page_sizes(a, g, total, in_size, out_size);
        atomicAdd(&hist[page_bucket(in_size, out_size)], 1u);
    }
    wave::sync();
    if (lane < kBuckets && hist[lane]) atomicAdd(a.status + kOrderHist + lane, hist[lane]);
}

__global__ void __launch_bounds__(64) brotlig_order_scatter_kernel(DecodeArgs a)
{
    __shared__ uint32_t cnt[kBuckets], base[kBuckets], start[kBuckets];
    const uint32_t lane = threadIdx.x, total = a.page_base[a.num_streams];
    if (a.order == nullptr || total > a.order_cap) return;
    const uint32_t mode = schedule_mode(a, total);
    if (mode == 0u) {                                                   // page order
        for (uint32_t g = blockIdx.x * 64u + lane; g < total; g += gridDim.x * 64u) a.order[g] = g;
        return;
    }
    {   // where each bucket starts in the schedule: exclusive prefix of the histogram, one bucket per lane
        const uint32_t h = lane < kBuckets ? a.status[kOrderHist + lane] : 0u;
        const uint32_t incl_half = wave::half_scan_incl(h);
        const uint32_t lower_total = wave::bcast(incl_half, 31u);
        const uint32_t incl = lane < 32u ? incl_half : incl_half + lower_total;
        if (lane < kBuckets) start[lane] = incl - h;
    }
    wave::sync();
    for (uint32_t g0 = blockIdx.x * 64u; g0 < total; g0 += gridDim.x * 64u) {      // uniform trip count
        const uint32_t g = g0 + lane;
        if (lane < kBuckets) cnt[lane] = 0u;
        wave::sync();
        uint32_t b = 0, rank = 0;
        if (g < total) {
            uint32_t in_size, out_size;
            page_sizes(a, g, total, in_size, out_size);
            b = page_bucket(in_size, out_size);
            rank = atomicAdd(&cnt[b], 1u);
        }
        wave::sync();
        if (lane < kBuckets) base[lane] = cnt[lane] ? atomicAdd(a.status + kOrderCursor + lane, cnt[lane]) : 0u;
        wave::sync();
        if (g < total) {
            const uint32_t p = start[b] + base[b] + rank;               // place in the schedule proper
            // folded: the front half of the schedule answers the even requests, the back half -- from the end -- the odd ones
            a.order[mode == 2u ? (p <= (total - 1u) >> 1 ? 2u * p : 2u * (total - 1u - p) + 1u) : p] = g;
        }
        wave::sync();
    }
}

// Pairing policy of the decode kernel (decode_pages): do neighbouring pages of the schedule differ in
// cost?  Up to 256 evenly spaced pairs (2k, 2k+1) are compared by compressed size; when more than a
// quarter of them differ by over 25 % (page kinds side by side) the two halves of a wavefront run free
// of each other, otherwise they stay in step (status word 3: the number of quarters of a page within
// which a free half waits for its neighbour -- 1 or 4).  One workgroup, after the order kernels.
__global__ void __launch_bounds__(64) brotlig_policy_kernel(DecodeArgs a)
{
    const uint32_t lane = threadIdx.x, total = a.page_base[a.num_streams];
    const bool ordered = a.order != nullptr && total <= a.order_cap;
    const uint32_t pairs = total / 2u, nsamp = min_u32(pairs, 256u);
    uint32_t differ = 0, valid = 0;
    const uint32_t stride = nsamp ? pairs / nsamp : 0u;
    for (uint32_t j = lane; j < nsamp; j += 64u) {
        const uint32_t g = 2u * (j * stride);                           // evenly spaced pairs (stride = pairs / nsamp, one exact division per wavefront)
        uint32_t sa, ua, sb, ub;
        page_sizes(a, ordered ? a.order[g] : g, total, sa, ua);
        page_sizes(a, ordered ? a.order[g + 1u] : g + 1u, total, sb, ub);
        const uint32_t big = sa > sb ? sa : sb, small = sa > sb ? sb : sa;
        ++valid;
        if ((big - small) * 4u > big) ++differ;
    }
    if (valid) atomicAdd(a.status + 3, differ | (valid << 16));
    wave::global_fence();
    wave::sync();
    if (lane == 0u) {
        const uint32_t packed = a.status[3];
        a.status[3] = (packed & 0xFFFFu) * 4u > (packed >> 16) ? 1u : 4u;
    }
}

}  // namespace brotlig
