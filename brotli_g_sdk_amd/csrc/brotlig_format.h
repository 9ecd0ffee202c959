// brotlig_format.h -- Brotli-G v1.1 container / page constants shared by the host C-ABI and the
// HIP kernels.  Values restate the reference headers (paths relative to the reference tree):
//   inc/DataStream.h:28-108          StreamHeader / PreconditionHeader bitfields
//   inc/common/BrotligConstants.h    alphabet sizes (:32-42), header widths (:47-68), limits (:77-90)
// Include after <hip/hip_runtime.h> (the functions are __host__ __device__).
#pragma once
#include <stdint.h>

namespace brotlig {

constexpr uint32_t kStreamId = 5;                 // BROTLIG_STREAM_ID
constexpr uint32_t kNumStreams = 32;              // sub-bitstreams per page (one per decode lane)
constexpr uint32_t kIcpAlphabet = 728;            // 704 commands + sentinel + 23 insert-only
constexpr uint32_t kDistAlphabet = 544;
constexpr uint32_t kLitAlphabet = 256;
constexpr uint32_t kSentinel = 704;
constexpr uint32_t kMinPageSize = 32768;
constexpr uint32_t kMaxPageSize = 131072;       // BROTLIG_MAX_PAGE_SIZE: what the reference's ENCODER stops at; the header's index 3 (256 KiB) decodes all the same
constexpr uint32_t kMaxSubBlocks = 6;
constexpr uint32_t kMaxMips = 32;

// Parsed view of the 8-byte stream header (+ optional 8-byte precondition header).
struct StreamInfo {
    uint32_t num_pages;
    uint32_t page_size;
    uint32_t last_page_size;     // 0 = last page is full
    uint32_t preconditioned;
    uint32_t header_bytes;       // 8 or 16: offset of the page table
    uint32_t precon_w0, precon_w1;
};

enum : uint32_t {           // bits of the batch status word (workspace word 0)
    kStatusBadHeader = 1u,  // magic / id check failed (src/BrotligDecoder.cpp:437-446)
    kStatusBadPage = 2u,    // a page failed a bounds check (the reference has none: undefined there)
};

// w0/w1 are the two little-endian dwords of the StreamHeader.
__host__ __device__ inline bool parse_stream_header(uint32_t w0, uint32_t w1, StreamInfo& s)
{
    const uint32_t id = w0 & 0xFF, magic = (w0 >> 8) & 0xFF;
    s.num_pages = w0 >> 16;
    s.page_size = kMinPageSize << (w1 & 3);
    s.last_page_size = (w1 >> 2) & 0x3FFFF;
    s.preconditioned = (w1 >> 20) & 1;
    s.header_bytes = s.preconditioned ? 16 : 8;
    s.precon_w0 = s.precon_w1 = 0;
    return id == (magic ^ 0xFF) && id == kStreamId;
}

__host__ __device__ inline uint32_t uncompressed_size(const StreamInfo& s)
{
    return s.num_pages * s.page_size - (s.last_page_size ? s.page_size - s.last_page_size : 0);
}

}  // namespace brotlig
