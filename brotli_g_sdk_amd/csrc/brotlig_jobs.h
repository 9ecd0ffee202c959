// brotlig_jobs.h -- a page's job: where its bytes are and where they go (page table walk, src/BrotligDecoder.cpp:310-314), and what the page schedule holds.
// Part of the gfx950 Brotli-G decode kernels; brotlig_kernels.h includes the parts in order and says what the whole replaces.
#pragma once
#include "brotlig_kernel_common.h"

namespace brotlig {

// -------------------------------------------------------------------------------------------
// Decode the pages `page_a` (lanes 0-31) and `page_b` (lanes 32-63); either may be absent.
struct PageJob {
    const uint8_t* in;      // compressed page
    uint32_t in_size;       // bytes
    uint32_t in_limit;      // bytes readable from `in` without leaving the input buffer
    uint8_t* out;           // where the page's bytes go (final output, or conditioned-space scratch)
    uint32_t out_size;
    uint32_t page_size;
    uint32_t page_off;      // offset of the page in its stream's (conditioned) byte space
    const DcTable* dc;      // non-null for preconditioned streams
    uint32_t stream;        // index of the page's stream in the batch
    bool     valid;
};

// dword of the compressed page at byte offset `rel`, zero beyond the readable input
__device__ __forceinline__ uint32_t br_load(const PageJob& job, uint32_t rel)
{
    return rel + 4u <= job.in_limit ? load_u32(job.in + rel) : 0u;
}

// byte k of the result = b0 + ... + bk (mod 256) of the dword's bytes
__device__ __forceinline__ uint32_t byte_prefix(uint32_t x)
{
    const uint32_t lo = (x & 0x00FF00FFu) * 0x00010001u;              // 16-bit fields (b0, b0+b2)
    const uint32_t hi = ((x >> 8) & 0x00FF00FFu) * 0x00010001u;       //               (b1, b1+b3)
    const uint32_t even = lo + (hi << 16), odd = lo + hi;             // (b0, b0+b1+b2), (b0+b1, b0+..+b3)
    return (even & 0x00FF00FFu) | ((odd & 0x00FF00FFu) << 8);
}
// adds the byte `c` to each byte of `x` (mod 256, no carries between bytes)
__device__ __forceinline__ uint32_t byte_add(uint32_t x, uint32_t c)
{
    const uint32_t cc = (c & 0xFFu) * 0x01010101u;
    return ((x & 0x7F7F7F7Fu) + (cc & 0x7F7F7F7Fu)) ^ ((x ^ cc) & 0x80808080u);
}

// A page of stream `s` failed: the batch-wide status word (the shader's meta[0], BrotliGCompute.hlsl:1757-1881) and the stream's own
// (round 5: a batch of up to 4 096 assets names the damaged ones).  Rare path, one lane.  `batch_word`: DecodeArgs::status for the page
// kernels; the schedule kernel collects in its own word (DecodeArgs::sync[kSyncStatus]) and publishes the batch word when it is done.
__device__ __forceinline__ void flag_bad_page(uint32_t* batch_word, const DecodeArgs& a, uint32_t s)
{
    atomicOr(batch_word, kStatusBadPage);
    atomicOr(&a.dc[s].status, kStatusBadPage);
}
__device__ __forceinline__ void flag_bad_page(const DecodeArgs& a, uint32_t s) { flag_bad_page(a.status, a, s); }

// What the schedule kernel writes into the page schedule (DecodeArgs::jobs, there whenever the workspace has room for it) for a batch of
// `total` pages -- the page kernel takes jobs[k] for its k-th request whatever it holds:
//   0  page order (jobs[k] = page k): the batch is too small for anything else to pay;
//   1  the schedule proper: bucket by bucket, dense pages first, similar pages side by side (large batches: every half-wave decodes many
//      pages, and two pages that share a wavefront cost the slower one's time in every phase of a round);
//   2  the schedule FOLDED (late round 5): the batch has more pages than the launch has wavefronts and at most twice as many -- every
//      half-wave gets one page at most, all at the start, and what the launch takes is its most loaded wavefront.  Even requests are answered
//      from the front of the schedule and odd ones from its back: the two halves of a wavefront ask together, so the densest page meets the
//      lightest, the second densest the second lightest ...  (4 096 textures with mip chains -- 6 827 pages of 64, 44 and 23 KiB -- in page
//      order: wavefronts with two full pages while others hold none; profiles/experiments/r05_many_textures.md.)  The wavefronts beyond
//      `total - waves` take one page each from the same counter, so from there on the parity of a request drifts: the pairing is exact for
//      the wavefronts that ask first (they hold the densest pages) and approximate after that.
#ifndef BROTLIG_TUNE_FOLD
#define BROTLIG_TUNE_FOLD 1
#endif
__device__ __forceinline__ uint32_t schedule_mode(const DecodeArgs& a, uint32_t total)
{
    const uint32_t waves = a.decode_waves;
    if (BROTLIG_TUNE_FOLD && waves != 0u && total > waves && total - waves <= waves) return 2u;
    return total >= 1024u * a.order_from_k ? 1u : 0u;
}

// How a walk reads the page prefix (DecodeArgs::page_base): with ordinary loads -- the page kernels, long after the schedule kernel wrote it;
// also a copy of it in LDS -- or, inside the schedule kernel, with loads that see what another workgroup has just written (brotlig_schedule.h).
struct PrefixPlain { const uint32_t* p; __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return p[i]; } };
struct PrefixCoherent { const uint32_t* p; __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return wave::agent_load_relaxed(p + i); } };

// Page `g` of the batch (pages counted stream after stream): stream lookup, page table walk (src/BrotligDecoder.cpp:310-314), bounds
// against the caller's buffers -- as the record the page kernels take.  A page that fails a check comes back
// without kJobValid; the caller says so to the status words.
template <class Prefix>
__device__ inline JobRecord walk_page(const DecodeArgs& a, const Prefix& page_base, uint32_t g)
{
    const StreamDesc* const streams = a.streams;
    const uint8_t* const in = a.in;
    const uint64_t in_bytes = a.in_bytes, out_bytes = a.out_bytes;
    // stream lookup: largest s with page_base[s] <= g
    uint32_t lo = 0, hi = a.num_streams;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (page_base(mid) <= g) lo = mid; else hi = mid; }
    const uint32_t i = g - page_base(lo);
    const uint64_t s_in = streams[lo].in_offset, s_out = streams[lo].out_offset;
    const uint8_t* sp = in + s_in;
    StreamInfo si;
    parse_stream_header(load_u32(sp), load_u32(sp + 4), si);
    const uint8_t* table = sp + si.header_bytes;
    const uint32_t np = si.num_pages;                       // (a stream the prepare phase refused has no pages in the prefix: no g leads here)
    const uint32_t off = i ? load_u32(table + 4u * i) : 0u;                                  // src/BrotligDecoder.cpp:310
    JobRecord r;
    r.in_size = i + 1u < np ? load_u32(table + 4u * (i + 1u)) - off : load_u32(table);       // :311
    const uint32_t out_size = (i + 1u == np && si.last_page_size) ? si.last_page_size : si.page_size;     // :314
    r.in_off = s_in + si.header_bytes + 4ull * np + off;
    const uint64_t in_end = stream_in_end(streams[lo], in_bytes);
    const uint64_t room = r.in_off < in_end ? in_end - r.in_off : 0;
    r.out_off = s_out + (uint64_t)i * si.page_size;
    r.stream = lo; r.page = i;
    // (an empty page is not a page: with a damaged table entry it can lie anywhere -- `room` is 0 beyond the stream and 0 > 0 let it through,
    // the bit readers then started at an address outside the input; found by the device soak of round 4)
    const bool ok = r.out_off + out_size <= stream_out_end(streams[lo], out_bytes) && r.in_size <= room && r.in_size != 0u &&
                    (si.preconditioned ? a.scratch : a.out) != nullptr;
    r.shape = out_size | ((msb_u32(si.page_size) - 15u) << 20) | (si.preconditioned ? kJobPrecon : 0u) | (ok ? kJobValid : 0u);
    return r;
}

// The record as the job a half-wave works on (meaningful when `ok`).
__device__ __forceinline__ PageJob job_of(const DecodeArgs& a, const JobRecord& r, bool ok)
{
    PageJob job;
    job.valid = ok && (r.shape & kJobValid) != 0u;
    job.in = a.in + r.in_off;
    job.in_size = r.in_size;
    // bytes readable from the page start: up to the end of the input buffer plus its 16 bytes of padding
    // (reads may run into the next stream: harmless, a valid page never consumes those bits)
    const uint64_t readable = r.in_off < a.in_bytes ? a.in_bytes - r.in_off + 16ull : 0ull;
    job.in_limit = (uint32_t)(readable > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : readable);
    const bool precon = (r.shape & kJobPrecon) != 0u;
    job.out = (precon ? a.scratch : a.out) + r.out_off;
    job.out_size = r.shape & 0x7FFFFu;
    job.page_size = kMinPageSize << ((r.shape >> 20) & 3u);
    job.page_off = r.page * job.page_size;
    job.dc = precon ? a.dc + r.stream : nullptr;
    job.stream = r.stream;
    return job;
}

// The job of the k-th request of the page counter (meaningful when `ok`): one 32-byte read of the schedule; without a schedule (a workspace
// of the minimum size) page k of the batch, walked from the page tables here.
__device__ inline PageJob fetch_job(const DecodeArgs& a, const JobRecord* jobs, uint32_t k, bool ok)
{
    JobRecord r;
    r.in_off = 0; r.in_size = 0; r.shape = 0; r.out_off = 0; r.stream = 0; r.page = 0;
    if (ok) {
        if (jobs != nullptr) r = jobs[k];
        else {
            r = walk_page(a, PrefixPlain{a.page_base}, k);
            if ((r.shape & kJobValid) == 0u) flag_bad_page(a, r.stream);
        }
    }
    return job_of(a, r, ok);
}
// a job that is none (the state of a half-wave before its first page)
__device__ __forceinline__ PageJob no_job(const DecodeArgs& a)
{
    JobRecord r;
    r.in_off = 0; r.in_size = 0; r.shape = 0; r.out_off = 0; r.stream = 0; r.page = 0;
    return job_of(a, r, false);
}

}  // namespace brotlig
