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
    uint32_t index;         // global page index (position in stream order, before the schedule)
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
// (round 5: a batch of up to 4 096 assets names the damaged ones).  Rare path, one lane.
__device__ __forceinline__ void flag_bad_page(const DecodeArgs& a, uint32_t s)
{
    atomicOr(a.status, kStatusBadPage);
    atomicOr(&a.dc[s].status, kStatusBadPage);
}

// What the order kernels write into the page schedule (DecodeArgs::order, there whenever the workspace has room for it) for a batch of
// `total` pages -- the page kernel takes order[k] for its k-th request whatever it holds:
//   0  page order (order[k] = k): the batch is too small for anything else to pay;
//   1  the schedule proper: bucket by bucket, dense pages first, similar pages side by side (large batches: every half-wave decodes many
//      pages, and two pages that share a wavefront cost the slower one's time in every phase of a round);
//   2  the schedule FOLDED (late round 5): the batch has more pages than the launch has wavefronts and at most twice as many -- every
//      half-wave gets one page at most, all at the start, and what the launch takes is its most loaded wavefront.  Even requests are answered
//      from the front of the schedule and odd ones from its back: the two halves of a wavefront ask together, so the densest page meets the
//      lightest, the second densest the second lightest ...  (4 096 textures with mip chains -- 6 827 pages of 64, 44 and 23 KiB -- in page
//      order: wavefronts with two full pages while others hold none; profiles/experiments/r05_many_textures.md.)
#ifndef BROTLIG_TUNE_FOLD
#define BROTLIG_TUNE_FOLD 1
#endif
__device__ __forceinline__ uint32_t schedule_mode(const DecodeArgs& a, uint32_t total)
{
    const uint32_t waves = a.decode_waves;
    if (BROTLIG_TUNE_FOLD && waves != 0u && total > waves && total - waves <= waves) return 2u;
    return total >= 1024u * a.order_from_k ? 1u : 0u;
}

// The job of global page index `g` (meaningful when `ok`): stream lookup, page table walk
// (src/BrotligDecoder.cpp:310-314), bounds against the caller's buffers.
__device__ inline PageJob fetch_job(const DecodeArgs& a, const uint32_t* order, uint32_t g, bool ok)
{
    PageJob job;
    job.valid = ok;
    job.in = nullptr; job.out = nullptr; job.in_size = job.out_size = 0; job.in_limit = 0; job.page_size = kMinPageSize;
    job.page_off = 0; job.dc = nullptr; job.index = 0; job.stream = 0;
    if (job.valid) {
        if (order != nullptr) g = order[g];                             // the schedule built by the order kernels
        job.index = g;
        const uint32_t* const page_base = a.page_base;
        const StreamDesc* const streams = a.streams;
        const uint8_t* const in = a.in;
        const uint64_t in_bytes = a.in_bytes, out_bytes = a.out_bytes;
        // stream lookup: largest s with page_base[s] <= g
        uint32_t lo = 0, hi = a.num_streams;
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (page_base[mid] <= g) lo = mid; else hi = mid; }
        const uint32_t i = g - page_base[lo];
        job.stream = lo;
        const uint64_t s_in = streams[lo].in_offset, s_out = streams[lo].out_offset;
        const uint8_t* sp = in + s_in;
        StreamInfo si;
        parse_stream_header(load_u32(sp), load_u32(sp + 4), si);
        const uint8_t* table = sp + si.header_bytes;
        const uint8_t* pages = table + 4u * si.num_pages;
        const uint32_t off = i ? load_u32(table + 4u * i) : 0u;                      // src/BrotligDecoder.cpp:310
        job.in_size = i + 1u < si.num_pages ? load_u32(table + 4u * (i + 1u)) - off : load_u32(table);   // :311
        job.out_size = (i + 1u == si.num_pages && si.last_page_size) ? si.last_page_size : si.page_size;  // :314
        job.page_size = si.page_size;
        job.in = pages + off;
        const uint64_t abs_in = (uint64_t)(job.in - in);
        const uint64_t in_end = stream_in_end(streams[lo], in_bytes);
        const uint64_t room = abs_in < in_end ? in_end - abs_in : 0;
        // bytes readable from the page start: up to the end of the input buffer plus its 16 bytes of padding
        // (reads may run into the next stream: harmless, a valid page never consumes those bits)
        const uint64_t readable = abs_in < in_bytes ? in_bytes - abs_in + 16ull : 0ull;
        job.in_limit = (uint32_t)(readable > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : readable);
        const uint64_t abs_out = s_out + (uint64_t)i * si.page_size;
        uint8_t* dst_base = si.preconditioned ? a.scratch : a.out;
        job.page_off = i * si.page_size;
        job.dc = si.preconditioned ? a.dc + lo : nullptr;
        job.out = dst_base + abs_out;
        // (an empty page is not a page: with a damaged table entry it can lie anywhere -- `room` is 0 beyond the stream and 0 > 0 let it through,
        // the bit readers then started at an address outside the input; found by the device soak of round 4)
        if (abs_out + job.out_size > stream_out_end(streams[lo], out_bytes) || job.in_size > room || job.in_size == 0u || dst_base == nullptr) {
            job.valid = false;
            flag_bad_page(a, lo);
        }
    }
    return job;
}

}  // namespace brotlig
