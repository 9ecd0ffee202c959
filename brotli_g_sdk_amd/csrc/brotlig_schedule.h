// brotlig_schedule.h -- the ONE kernel in front of the page decode: stream headers -> page counts -> prefix, the page schedule as job records,
// the pairing policy.
// Part of the gfx950 Brotli-G decode kernels; brotlig_kernels.h includes the parts in order and says what the whole replaces.
//
// What it replaces: K1 of the reference shader (the stream queue walk, BrotliGCompute.hlsl:1757-1881) and the per-page table walk of its
// page loop (:1823-1834; CPU twin src/BrotligDecoder.cpp:296-329).  Rounds 1-5 spread this over FIVE launches in front of every decode
// (memset, prepare [+ finish], order count, order scatter, policy: 76 us of the benchmark's 8.4 ms step, 42 us of config 2's 0.55 ms), each
// waiting for the one before through the stream.  Round 6: one kernel whose workgroups wait for each other.
//
// How workgroups of one kernel may wait for each other without a grid barrier: every workgroup takes a TICKET from a counter when it starts,
// and the work is laid out as phases in ticket order -- prepare, scan, finalize, count, scatter; whoever finishes the last phase last
// computes the pairing policy and publishes the batch.  A workgroup working on an item of
// phase k waits (polls a counter of finished items) until phase k - 1 is complete.  Everything it waits for belongs to LOWER tickets, i.e. to
// workgroups that started before it and therefore hold their execution slots: no cycle, whatever else shares the device and however few
// of the kernel's workgroups are resident at a time (the argument of decoupled look-back scans).  The simulator (tests/sim) runs the
// workgroups one after the other in ticket order, where no wait ever has to wait.
//
// No memset in front of it either.  The kernel's own words (tickets, phase counters) are left at zero by the last item of every launch, and
// a 64-bit cookie beside them says so: a workgroup that finds the cookie goes straight to its ticket -- one atomic add.  A workgroup that
// does not (a fresh workspace, whatever it holds) competes for the right to initialise -- one compare-and-swap on the cookie, marked with
// the launch's tag (DecodeArgs::launch_tag) so that the mark of a launch that died is not mistaken for a live one --, the winner zeroes the
// words and writes the cookie, the others wait for it: once per workspace.  (Garbage that happens to BE the cookie: 2^-64 per allocation.
// Round 6's first version took every ticket by compare-and-swap under a per-launch tag: a thousand workgroups retrying against each other
// -- half a million atomics, 3.8 ms in front of an 8 ms decode.)  Until the policy item has run, the batch's page count reads 0 and its
// status "failed": a schedule that did not complete decodes nothing and says so; a workgroup that gives up waiting (it never should)
// also takes the cookie away, so that the next launch starts from scratch.
#pragma once
#include "brotlig_jobs.h"
#include "brotlig_decondition.h"

namespace brotlig {

// -------------------------------------------------------------------------------------------
// Page schedule.  The decode kernel runs two pages per wavefront and pays the maximum of the two in
// every phase of a round, so it matters which pages meet: the same 4 GiB of mixed pages decode 12 %
// faster when similar pages are neighbours.  Pages are therefore grouped into buckets by
// compressed size relative to the page size (an eighth of an octave per bucket since round 5, see below; stored pages last) and
// handed out bucket by bucket, dense pages first (they are the slow ones, which also shortens the
// tail of the launch).  Two passes over the page tables: count, then scatter into `jobs`.

// compressed and decompressed size of global page g (the walk of walk_page without the bounds)
template <class Prefix>
__device__ inline void page_sizes(const DecodeArgs& a, const Prefix& page_base, uint32_t g, uint32_t total, uint32_t& in_size, uint32_t& out_size)
{
    uint32_t lo = 0, hi = a.num_streams;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (page_base(mid) <= g) lo = mid; else hi = mid; }
    const uint32_t first = page_base(lo);
    const uint32_t i = g - first;
    const uint32_t np = (lo + 1u < a.num_streams ? page_base(lo + 1u) : total) - first;
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
    uint32_t b = 0, t = (uint32_t)(((uint64_t)out_size * kBucketStep16) >> 16);     // out_size <= 256 KiB
    while (b < kBuckets - 2u && in_size <= t) { ++b; t = (uint32_t)(((uint64_t)t * kBucketStep16) >> 16); }
    return b;
}
constexpr uint32_t kOrderHist = 8, kOrderCursor = 8 + kBuckets;         // status word offsets
constexpr uint32_t kStatusWords = 8u + 2u * 64u;                        // the workspace header: room for the widest setting
static_assert(kOrderCursor + kBuckets <= kStatusWords && kBuckets <= 64u, "status words; one lane per bucket in the schedule kernel");
constexpr uint32_t kStatusPages = 6;                                    // status word: pages of the batch (the schedule kernel's own copy)

// ---- phases and their items ------------------------------------------------------------------------------------------
enum : uint32_t { kPhasePrepare, kPhaseScan, kPhaseFinalize, kPhaseCount, kPhaseScatter, kPhasePolicy, kSchedPhases };
static_assert(kSyncDone + kSchedPhases <= kSyncStatus && kSyncStatus < kSyncWords, "the schedule kernel's words");
constexpr uint32_t kSchedThreads = 512;                                 // threads of a workgroup of the schedule kernel: pages a count / scatter item walks per step
constexpr uint32_t kSchedTimedTickets = 4096;                           // diagnostics: tickets whose four time stamps fit the profile buffer
struct SchedShape { uint32_t items[kSchedPhases]; };
// `workers`: items of the count and of the scatter phase (each walks every workers-th group of kSchedThreads pages).
__host__ __device__ inline SchedShape sched_shape(uint32_t num_streams, uint32_t workers)
{
    const uint32_t P = (num_streams + 63u) / 64u;                       // chunks of 64 streams: one wavefront's work in the prepare and finalize phases
    const uint32_t per = kSchedThreads / 64u;                           // ... and a workgroup has this many wavefronts
    SchedShape s;
    s.items[kPhasePrepare] = (P + per - 1u) / per;
    s.items[kPhaseScan] = P > 1u ? 1u : 0u;                             // (one chunk: its prepare item is the scan as well)
    s.items[kPhaseFinalize] = P > 1u ? (P + per - 1u) / per : 0u;
    s.items[kPhaseCount] = workers; s.items[kPhaseScatter] = workers;
    s.items[kPhasePolicy] = 0u;                                         // (the scatter item that finishes LAST runs the policy: see the kernel)
    return s;
}
__host__ __device__ inline uint32_t sched_grid(uint32_t num_streams, uint32_t workers)
{
    const SchedShape s = sched_shape(num_streams, workers);
    uint32_t g = 0;
    for (uint32_t k = 0; k < kSchedPhases; ++k) g += s.items[k];
    return g;
}
__host__ __device__ inline uint32_t sched_workers_of_grid(uint32_t num_streams, uint32_t grid)     // inverse of sched_grid
{
    const uint32_t fixed = sched_grid(num_streams, 0u);
    return grid > fixed ? (grid - fixed) / 2u : 0u;
}
// how often a workgroup looks at a counter before it gives up (the batch then fails as a whole, see the head of the file)
#ifndef BROTLIG_SCHED_SPIN_LIMIT
#define BROTLIG_SCHED_SPIN_LIMIT (1u << 23)
#endif

// inclusive prefix sum over the 64 lanes of the workgroup, and the total
__device__ __forceinline__ uint32_t sched_scan64(uint32_t v, uint32_t lane, uint32_t& total)
{
    const uint32_t lo = wave::half_scan_incl(v);
    const uint32_t lo_total = wave::half_bcast(lo, 31);
    const uint32_t first = wave::bcast(lo_total, 0), second = wave::bcast(lo_total, 32);
    total = first + second;
    return lane < 32u ? lo : lo + first;
}

// Words that cross workgroups inside this kernel -- the page prefix, the shared line of the stream records (DcTable), the status words --
// are written with write-through stores and read with loads that look beyond the own L2 (wave::agent_*, brotlig_wave_ops.h).  Everything
// else (the pre-conditioning tables, the job records) is written with ordinary stores and read by LATER kernels only.
__device__ __forceinline__ void put(uint32_t* p, uint32_t v) { wave::agent_store_relaxed(p, v); }
__device__ __forceinline__ uint32_t get(const uint32_t* p) { return wave::agent_load_relaxed(p); }

// ---- phase "prepare", item c: the streams 64 c .. 64 c + 63.  Validates headers (src/BrotligDecoder.cpp:437-446; the page table must lie
// inside the input), page counts -> prefix INSIDE the chunk, pre-conditioning tables (BrotligDataconditionParams::Initialize,
// inc/common/BrotligDataConditioner.h:92-237), the chunk's totals.  (Rounds 1-4 walked all streams in ONE workgroup, 64 per step, every
// step a chain of dependent loads -- descriptor, header, table: 1.3 ms for a batch of 65 536 small streams, a quarter of its whole
// decode; round 5, profiles/experiments/r05_many_streams.md.)  Returns the kStatus* bits it found (the same in every lane).
__device__ inline uint32_t sched_prepare(const DecodeArgs& a, uint32_t c, uint32_t lane, uint32_t& chunk_pages, uint32_t& chunk_supers, uint32_t& chunk_precon)
{
    const uint32_t s = c * 64u + lane;
    uint32_t pages = 0, supers = 0;
    bool bad = false, precon = false;
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
        else bad = true;
        DcTable& t = a.dc[s];
        t.precon = 0;
        uint32_t st = ok ? 0u : kStatusBadHeader;                   // the stream's own status word (pages add kStatusBadPage)
        if (pages && si.preconditioned) {
            // the texture described by the precondition header is the stream's output (:478): the de-conditioning
            // kernel writes all of it, whatever happened to the stream's pages
            if (a.scratch == nullptr || usz > 0xFFFFFFFFull || !dc_init(t, load_u32(p + 8), load_u32(p + 12), (uint32_t)usz)) {
                t.precon = 0; pages = 0;                            // the reference has undefined behaviour here
                st = kStatusBadHeader;
                bad = true;
            } else { precon = true; supers = t.item_prefix[t.num_mips] >> 8; }
        }
        put(&t.status, st);
    }
    const uint32_t incl = sched_scan64(pages, lane, chunk_pages);
    const uint32_t su_incl = sched_scan64(supers, lane, chunk_supers);      // the same for the de-conditioning super-tiles
    chunk_precon = (uint32_t)__popcll(wave::ballot64(precon));
    if (s < a.num_streams) { put(a.page_base + s, incl - pages); put(&a.dc[s].super_base, su_incl - supers); }
    if (lane == 0u) { DcTable& t0 = a.dc[c * 64u]; put(&t0.chunk_pages, chunk_pages); put(&t0.chunk_supers, chunk_supers); put(&t0.chunk_precon, chunk_precon); }
    return wave::any(bad) ? kStatusBadHeader : 0u;
}

// the batch's totals, and the schedule's histogram and cursors back to zero (the end of the scan phase; of the prepare item when there is one chunk)
__device__ inline void sched_publish_totals(const DecodeArgs& a, uint32_t pages, uint32_t supers, uint32_t precon, uint32_t lane)
{
    for (uint32_t w = lane; w < 2u * kBuckets; w += 64u) put(a.status + kOrderHist + w, 0u);
    if (lane == 0u) { put(a.status + kStatusPages, pages); put(a.status + 5, supers); put(a.status + 2, precon); }
}

// ---- phase "scan" (one item, batches of more than 64 streams): the chunks' totals -> what lies before each chunk; the batch's totals.
__device__ inline void sched_scan(const DecodeArgs& a, uint32_t lane)
{
    const uint32_t P = (a.num_streams + 63u) / 64u;
    uint32_t run_pages = 0, run_supers = 0, run_precon = 0;
    for (uint32_t base = 0; base < P; base += 64u) {                    // (uniform trip count)
        const uint32_t c = base + lane;
        const bool in = c < P;
        DcTable& t = a.dc[(in ? c : 0u) * 64u];
        const uint32_t vp = in ? get(&t.chunk_pages) : 0u, vs = in ? get(&t.chunk_supers) : 0u, vc = in ? get(&t.chunk_precon) : 0u;
        uint32_t tp, ts, tc;
        const uint32_t ip = sched_scan64(vp, lane, tp), is = sched_scan64(vs, lane, ts);
        (void)sched_scan64(vc, lane, tc);
        if (in) { put(&t.chunk_pages_before, run_pages + ip - vp); put(&t.chunk_supers_before, run_supers + is - vs); }
        run_pages += tp; run_supers += ts; run_precon += tc;
    }
    sched_publish_totals(a, run_pages, run_supers, run_precon, lane);
}

// ---- phase "finalize", chunk c (batches of more than 64 streams; one wavefront): the prefix inside the chunk becomes the batch's.
__device__ inline void sched_finalize(const DecodeArgs& a, uint32_t c, uint32_t lane)
{
    const uint32_t before = get(&a.dc[c * 64u].chunk_pages_before), before_su = get(&a.dc[c * 64u].chunk_supers_before);
    const uint32_t s = c * 64u + lane;
    if (s < a.num_streams) { put(a.page_base + s, get(a.page_base + s) + before); put(&a.dc[s].super_base, get(&a.dc[s].super_base) + before_su); }
}

// The page prefix as the count and scatter items read it: a batch of up to 64 streams has its 64 words copied into LDS once per workgroup
// (one coherent load per lane of the first wavefront; the stream lookups of the walks then stay on chip), a larger one is searched where it lies.
struct SchedPrefix {
    const uint32_t* lds; const uint32_t* global;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return lds != nullptr ? lds[i] : wave::agent_load_relaxed(global + i); }
};
// (every wavefront of the workgroup calls it: a workgroup barrier inside)
__device__ inline SchedPrefix sched_prefix(const DecodeArgs& a, uint32_t tid, uint32_t* lds64)
{
    if (a.num_streams > 64u) return SchedPrefix{nullptr, a.page_base};
    if (tid < 64u) lds64[tid] = tid < a.num_streams ? get(a.page_base + tid) : 0xFFFFFFFFu;
    __syncthreads();
    return SchedPrefix{lds64, a.page_base};
}

// ---- phase "count", item i of `workers` (the whole workgroup: kSchedThreads pages per step): pages per bucket (only a batch that gets the
// schedule proper, or the folded one, needs them)
__device__ inline void sched_count(const DecodeArgs& a, uint32_t i, uint32_t workers, uint32_t total, uint32_t tid, uint32_t* hist, uint32_t* lds64)
{
    if (a.jobs == nullptr || total > a.jobs_cap || schedule_mode(a, total) == 0u) return;
    if (tid < kBuckets) hist[tid] = 0u;
    const SchedPrefix prefix = sched_prefix(a, tid, lds64);
    __syncthreads();
    for (uint32_t g = i * kSchedThreads + tid; g < total; g += workers * kSchedThreads) {
        uint32_t in_size, out_size;
        page_sizes(a, prefix, g, total, in_size, out_size);
        atomicAdd(&hist[page_bucket(in_size, out_size)], 1u);
    }
    __syncthreads();
    if (tid < kBuckets && hist[tid]) wave::agent_add_relaxed(a.status + kOrderHist + tid, hist[tid]);
}

// ---- phase "scatter", item i of `workers` (the whole workgroup): every page of its share walked once (walk_page) and its record put where
// the schedule wants it
__device__ inline void sched_scatter(const DecodeArgs& a, uint32_t i, uint32_t workers, uint32_t total, uint32_t tid, uint32_t* cnt, uint32_t* base, uint32_t* start, uint32_t* lds64)
{
    if (a.jobs == nullptr || total > a.jobs_cap) return;               // no room for a schedule: the page kernels walk the tables themselves
    uint32_t* const bad_word = a.sync + kSyncStatus;
    const uint32_t mode = schedule_mode(a, total);
    const SchedPrefix prefix = sched_prefix(a, tid, lds64);
    if (mode == 0u) {                                                   // page order
        for (uint32_t g = i * kSchedThreads + tid; g < total; g += workers * kSchedThreads) {
            const JobRecord r = walk_page(a, prefix, g);
            if ((r.shape & kJobValid) == 0u) flag_bad_page(bad_word, a, r.stream);
            a.jobs[g] = r;
        }
        return;
    }
    if (tid < 64u) {   // where each bucket starts in the schedule: exclusive prefix of the histogram, one bucket per lane of the first wavefront
        const uint32_t h = tid < kBuckets ? get(a.status + kOrderHist + tid) : 0u;
        uint32_t all;
        const uint32_t incl = sched_scan64(h, tid, all);
        if (tid < kBuckets) start[tid] = incl - h;
    }
    __syncthreads();
    for (uint32_t g0 = i * kSchedThreads; g0 < total; g0 += workers * kSchedThreads) {      // uniform trip count
        const uint32_t g = g0 + tid;
        if (tid < kBuckets) cnt[tid] = 0u;
        __syncthreads();
        uint32_t b = 0, rank = 0;
        JobRecord r;
        r.in_off = 0; r.in_size = 0; r.shape = 0; r.out_off = 0; r.stream = 0; r.page = 0;
        if (g < total) {
            r = walk_page(a, prefix, g);
            if ((r.shape & kJobValid) == 0u) flag_bad_page(bad_word, a, r.stream);
            b = page_bucket(r.in_size, r.shape & 0x7FFFFu);
            rank = atomicAdd(&cnt[b], 1u);
        }
        __syncthreads();
        if (tid < kBuckets) base[tid] = cnt[tid] ? wave::agent_add_relaxed(a.status + kOrderCursor + tid, cnt[tid]) : 0u;
        __syncthreads();
        if (g < total) {
            const uint32_t p = start[b] + base[b] + rank;               // place in the schedule proper
            // folded: the front half of the schedule answers the even requests, the back half -- from the end -- the odd ones
            a.jobs[mode == 2u ? (p <= (total - 1u) >> 1 ? 2u * p : 2u * (total - 1u - p) + 1u) : p] = r;
        }
        __syncthreads();
    }
}

// ---- the policy (run by the scatter item that finishes last): the pairing policy of the decode kernel (decode_pages) -- do pages that share a wavefront differ
// in cost?  Status word 3: the number of quarters of a page within which a free half waits for its neighbour -- 4: the halves stay in step
// (neighbours are alike), 1: they run free of each other; 0 for a batch in which no two pages can meet.
//   * pages in stream order (no schedule, or schedule_mode 0): up to 64 evenly spaced pairs (2k, 2k+1) are compared by compressed size, read
//     from the page tables; more than a quarter of them more than 25 % apart -> 1.  (256 pairs until round 6: four dependent walks in a row on
//     the one item every page kernel waits for.)
//   * the schedule proper: neighbours lie in one bucket (an eighth of an octave: 9 %) or on one of at most 63 bucket boundaries -> 4, without
//     looking.  (Rounds 1-5 sampled the schedule and found just that.)
//   * the folded schedule pairs the k-th page from the front with the k-th from the back: the same 256 samples from the HISTOGRAM -- the
//     buckets of ranks k and total - 1 - k four or more apart (2^(-4/8) = 0.71 < 0.75).
// Then the batch is PUBLISHED: page counter, status word, page count -- the page kernels, launched behind this one, start from these.
__device__ inline void sched_policy(const DecodeArgs& a, uint32_t total, uint32_t lane, uint32_t* lds64)
{
    uint32_t policy = 0u;
    if (a.may_pair != 0u) {
        const bool scheduled = a.jobs != nullptr && total <= a.jobs_cap;
        const uint32_t mode = scheduled ? schedule_mode(a, total) : 0u;
        const uint32_t pairs = total / 2u, nsamp = min_u32(pairs, mode == 0u ? 64u : 256u);      // (page order: one pair per lane, read from the page tables)
        const uint32_t stride = nsamp ? pairs / nsamp : 0u;             // (one exact division per wavefront)
        uint32_t differ = 0;
        if (mode == 0u) {
            // (this item is one wavefront: its own copy of the prefix, no workgroup barrier)
            if (a.num_streams <= 64u) { lds64[lane] = lane < a.num_streams ? get(a.page_base + lane) : 0xFFFFFFFFu; wave::sync(); }
            const SchedPrefix prefix{a.num_streams <= 64u ? lds64 : nullptr, a.page_base};
            for (uint32_t j = lane; j < nsamp; j += 64u) {
                const uint32_t g = 2u * (j * stride);                   // evenly spaced pairs
                uint32_t sa, ua, sb, ub;
                page_sizes(a, prefix, g, total, sa, ua);
                page_sizes(a, prefix, g + 1u, total, sb, ub);
                const uint32_t big = sa > sb ? sa : sb, small = sa > sb ? sb : sa;
                if ((big - small) * 4u > big) ++differ;
            }
        } else if (mode == 2u) {
            // inclusive prefix of the histogram in LDS: the bucket of rank r is the first one whose prefix exceeds r
            const uint32_t h = lane < kBuckets ? get(a.status + kOrderHist + lane) : 0u;
            uint32_t all;
            lds64[lane] = sched_scan64(h, lane, all);
            wave::sync();
            for (uint32_t j = lane; j < nsamp; j += 64u) {
                const uint32_t k = j * stride, front = k, back = total - 1u - k;
                uint32_t bf = 0, bb = 0;
                for (uint32_t q = 0; q < kBuckets; ++q) { bf += lds64[q] <= front ? 1u : 0u; bb += lds64[q] <= back ? 1u : 0u; }
                if (bb >= bf + 4u) ++differ;
            }
        }
        uint32_t all;
        (void)sched_scan64(differ, lane, all);
        policy = all * 4u > nsamp ? 1u : 4u;
    }
    if (lane == 0u) {
        put(a.status + 3, policy);
        put(a.work_counter, 0u);
        put(a.status, get(a.sync + kSyncStatus));                       // what the prepare and scatter items found; the page kernels add theirs
        put(a.sync + kSyncStatus, 0u);                                  // (clean for the next launch)
        put(a.page_base + a.num_streams, total);
    }
}

// ---- tickets ------------------------------------------------------------------------------------------------------------
constexpr uint64_t kSchedMagic = 0xB407116A5C4ED01Eull;                // the cookie: "the words are clean"
constexpr uint64_t kSchedInit = 0x1B17A11500000000ull;                  // | low 32 bits of the launch tag: "being initialised by a workgroup of that launch"
// Takes a ticket (one lane; see the head of the file).  False: gave up waiting for the initialisation.
__device__ inline bool sched_take_ticket(const DecodeArgs& a, uint32_t& ticket)
{
    uint64_t* const cookie = reinterpret_cast<uint64_t*>(a.sync + kSyncCookie);
    uint64_t c = wave::agent_load_relaxed64(cookie);
    if (c != kSchedMagic) {
        const uint64_t mine = kSchedInit | (a.launch_tag & 0xFFFFFFFFull);
        bool winner = false;
        while (!winner && c != kSchedMagic && c != mine) winner = wave::agent_cas64(cookie, c, mine);      // (a failed attempt leaves what it found in c)
        if (winner) {
            for (uint32_t k = kSyncTicket; k < kSyncWords; ++k) put(a.sync + k, 0u);
            wave::lane_stores_done();
            wave::agent_store_relaxed64(cookie, kSchedMagic);
        } else {
            for (uint32_t spins = 0; wave::agent_load_relaxed64(cookie) != kSchedMagic; ++spins) {
                if (spins >= BROTLIG_SCHED_SPIN_LIMIT) return false;
                wave::long_nap();
            }
        }
    }
    ticket = wave::agent_add_relaxed(a.sync + kSyncTicket, 1u);
    return true;
}
// the launch cannot complete: the batch has no pages and says "failed", the next launch on this workspace initialises again
__device__ inline void sched_give_up(const DecodeArgs& a, uint32_t tid)
{
    if (tid == 0u) {
        put(a.page_base + a.num_streams, 0u);
        atomicOr(a.status, kStatusBadPage);
        wave::agent_store_relaxed64(reinterpret_cast<uint64_t*>(a.sync + kSyncCookie), 0ull);
    }
}
__device__ inline bool sched_wait_phase(const DecodeArgs& a, uint32_t phase, uint32_t items)
{
    // (one wavefront per workgroup looks, a few hundred at most; a look is a trip to memory, ~1 us: a short nap between two is enough)
    for (uint32_t spins = 0; get(a.sync + kSyncDone + phase) < items; ++spins) {
        if (spins >= BROTLIG_SCHED_SPIN_LIMIT) return false;
        wave::long_nap();
    }
    wave::stores_done();                                                // (nothing of mine moves in front of what I waited for)
    return true;
}

// Grid: 1 workgroup for a small batch (host: up to 64 streams and kSchedSmallPages pages) -- the phases one after the other, no ticket --
// or sched_grid(streams, workers) workgroups.  Workgroups of kSchedThreads threads: the count and scatter items walk that many pages per step
// with all their wavefronts; every other item is the work of the first wavefront.  (Workgroups of ONE wavefront, round 6's first form: the
// device starts such workgroups at 10 .. 40 per microsecond -- a thousand of them took 25 us to arrive -- and each paid its own ticket,
// its own looks at the phase counters and its own 64 atomics on the histogram, all on one cache line: 90 us for 65 536 pages.)
__global__ void __launch_bounds__(kSchedThreads) brotlig_schedule_kernel(DecodeArgs a)
{
    __shared__ uint32_t lds[3u * kBuckets + 64u + 2u];
    uint32_t* const lds64 = lds + 3u * kBuckets;
    uint32_t* const ctl = lds64 + 64u;                                  // [0] the workgroup's ticket, [1] "go on"
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const bool first = tid < 64u;                                       // the first wavefront
    if (gridDim.x == 1u) {
        if (tid == 0u) { put(a.sync + kSyncStatus, 0u); put(a.page_base + a.num_streams, 0u); }
        if (first) {
            wave::stores_done();
            uint32_t pages, supers, precon;
            const uint32_t bad = sched_prepare(a, 0u, lane, pages, supers, precon);
            if (lane == 0u && bad) atomicOr(a.sync + kSyncStatus, bad);
            sched_publish_totals(a, pages, supers, precon, lane);
            if (lane == 0u) ctl[0] = pages;
            wave::stores_done();
        }
        __syncthreads();
        const uint32_t pages = ctl[0];
        sched_count(a, 0u, 1u, pages, tid, lds, lds64);
        wave::stores_done();
        __syncthreads();
        sched_scatter(a, 0u, 1u, pages, tid, lds, lds + kBuckets, lds + 2u * kBuckets, lds64);
        wave::stores_done();
        __syncthreads();
        if (first) sched_policy(a, pages, lane, lds64);
        return;
    }
    const uint32_t workers = sched_workers_of_grid(a.num_streams, gridDim.x);
    const SchedShape shape = sched_shape(a.num_streams, workers);
    // diagnostics (BrotligDecodePhaseProfile only: DecodeArgs::prof is null in every other launch): when each workgroup came, got its ticket,
    // saw its phase open and was done -- four ticks of the 100 MHz counter per ticket, behind the page kernel's own records
    unsigned long long* const times = a.prof != nullptr ? a.prof + kNumPhases + 2u * a.decode_waves : nullptr;
    const unsigned long long t_came = times != nullptr ? wave::realtime() : 0ull;
    if (tid == 0u) { uint32_t t0 = 0; ctl[1] = sched_take_ticket(a, t0) ? 1u : 0u; ctl[0] = t0; }
    __syncthreads();
    const uint32_t t = ctl[0];
    if (ctl[1] == 0u) { sched_give_up(a, tid); return; }
    // first of this launch: the batch is "not there yet" (see the head of the file)
    if (t == 0u && tid == 0u) { put(a.status, kStatusBadPage); put(a.page_base + a.num_streams, 0u); }
    uint32_t phase = 0, item = t;
    while (phase < kSchedPhases && item >= shape.items[phase]) { item -= shape.items[phase]; ++phase; }
    if (phase >= kSchedPhases) return;                                  // (a surplus workgroup)
    const unsigned long long t_ticket = times != nullptr ? wave::realtime() : 0ull;
    if (phase > 0u) {
        // everything before my phase is complete when the last phase before it that has items is (each of ITS items waited the same way);
        // the first wavefront looks, the others wait for it at the barrier
        uint32_t prev = phase - 1u;
        while (shape.items[prev] == 0u) --prev;                         // (the prepare phase always has items)
        __syncthreads();                                                // (ctl[1] was read by everybody)
        if (first) {
            bool ok;
            if (phase == kPhaseScatter) {
                // a schedule in page order needs no histogram: its scatter items only wait for what the count items waited for
                uint32_t before = kPhaseCount - 1u;
                while (shape.items[before] == 0u) --before;
                ok = sched_wait_phase(a, before, shape.items[before]);
                const uint32_t pages = get(a.status + kStatusPages);
                const bool page_order = a.jobs == nullptr || pages > a.jobs_cap || schedule_mode(a, pages) == 0u;
                if (ok && !page_order) ok = sched_wait_phase(a, prev, shape.items[prev]);
            } else ok = sched_wait_phase(a, prev, shape.items[prev]);
            if (lane == 0u) ctl[1] = ok ? 1u : 0u;
        }
        __syncthreads();
        if (ctl[1] == 0u) { sched_give_up(a, tid); return; }
    }
    const unsigned long long t_open = times != nullptr ? wave::realtime() : 0ull;
    const uint32_t total = phase > kPhaseFinalize ? get(a.status + kStatusPages) : 0u;
    switch (phase) {
    case kPhasePrepare: {
        // every wavefront of the workgroup one chunk of 64 streams (a batch of 65 536 small streams: 128 workgroups instead of 1 024 that
        // the device starts one by one, 50 ns apiece)
        const uint32_t c = item * (kSchedThreads / 64u) + (tid >> 6);
        if (c * 64u < a.num_streams) {
            uint32_t pages, supers, precon;
            const uint32_t found = sched_prepare(a, c, lane, pages, supers, precon);
            if (lane == 0u && found) atomicOr(a.sync + kSyncStatus, found);
            if (shape.items[kPhaseScan] == 0u) sched_publish_totals(a, pages, supers, precon, lane);     // one chunk: its totals are the batch's
        }
        break;
    }
    case kPhaseScan: if (first) sched_scan(a, lane); break;
    case kPhaseFinalize: {
        const uint32_t c = item * (kSchedThreads / 64u) + (tid >> 6);
        if (c * 64u < a.num_streams) sched_finalize(a, c, lane);
        break;
    }
    case kPhaseCount: sched_count(a, item, workers, total, tid, lds, lds64); break;
    case kPhaseScatter: sched_scatter(a, item, workers, total, tid, lds, lds + kBuckets, lds + 2u * kBuckets, lds64); break;
    default: break;
    }
    // my item is done: every wavefront's stores have arrived, then the count
    wave::stores_done();
    __syncthreads();
    if (tid == 0u) {
        if (times != nullptr && t < kSchedTimedTickets) {
            times[4u * t] = t_came; times[4u * t + 1u] = t_ticket; times[4u * t + 2u] = t_open; times[4u * t + 3u] = wave::realtime();
        }
        const uint32_t before = wave::agent_add_relaxed(a.sync + kSyncDone + phase, 1u);
        ctl[0] = (phase == kPhaseScatter && before + 1u == shape.items[kPhaseScatter]) ? 1u : 0u;
    }
    __syncthreads();
    if (ctl[0] == 0u || !first) return;
    // The scatter item that finished LAST: every record is in place (the others' stores had arrived before they counted).  It runs the policy,
    // publishes the batch and leaves the kernel's words clean for the next launch -- after it has seen every count item finish as well (a
    // schedule in page order does not make its scatter items wait for them), so that nobody counts into a word that was already cleaned.
    if (!sched_wait_phase(a, kPhaseCount, shape.items[kPhaseCount])) { sched_give_up(a, tid); return; }
    sched_policy(a, total, lane, lds64);
    wave::stores_done();
    if (lane == 0u) {
        put(a.sync + kSyncTicket, 0u);
        for (uint32_t k = 0; k < kSchedPhases; ++k) put(a.sync + kSyncDone + k, 0u);
        if (times != nullptr) times[4u * kSchedTimedTickets - 1u] = wave::realtime();       // (when the batch was published)
    }
}

}  // namespace brotlig
