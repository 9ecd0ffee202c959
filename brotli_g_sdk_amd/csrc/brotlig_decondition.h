// brotlig_decondition.h -- pre-conditioning tables (inc/common/BrotligDataConditioner.h:92-237) and the de-conditioning kernel (PageDecoder.cpp:243-265, :406-444).
// Part of the gfx950 Brotli-G decode kernels; brotlig_kernels.h includes the parts in order and says what the whole replaces.
#pragma once
#include "brotlig_kernel_common.h"

namespace brotlig {

// NARROW mips (late round 5).  Under the 2 x 2 swizzle (PageDecoder.cpp:416-436) a pair of texture rows is 2 W consecutive blocks of every
// conditioned sub-stream, whatever W: a mip narrower than a super-tile's 128 columns -- W a power of two, H even, rows without padding -- is
// cut into super-tiles of 128 / W row pairs instead of one, each again 256 CONSECUTIVE blocks per sub-stream (the wide path's shape; before,
// such mips went through the per-block gather, one half-empty super-tile per row pair).  Returns log2 of the row pairs per super-tile
// (0: one, the general case).  dc_init counts the super-tiles with it, dc_texture walks them with it.
__device__ __forceinline__ uint32_t dc_row_group_log2(uint32_t W, uint32_t H, uint32_t pitch, uint32_t bb, uint32_t swizzle)
{
    const bool narrow = swizzle != 0u && W >= 2u && W < 128u && (W & (W - 1u)) == 0u && (H & 1u) == 0u && pitch == W * bb;
    return narrow ? 7u - (31u - (uint32_t)__clz((int)W)) : 0u;
}

// -------------------------------------------------------------------------------------------
// Pre-conditioning tables (inc/common/BrotligDataConditioner.h:92-237).  `w0`/`w1` are the two
// dwords of the PreconditionHeader; `out_size` is the stream's decompressed size, which must equal
// the texture size (:219).
__device__ inline bool dc_init(DcTable& t, uint32_t w0, uint32_t w1, uint32_t out_size)
{
    const uint32_t fmt = w1 & 0xFFu;
    // sub-block sizes per format, 4 bits each, first sub-block in the low nibble (:98-175)
    const uint32_t sizes = fmt == 1u ? 0x422u : fmt == 2u ? 0x4228u : fmt == 3u ? 0x422611u
                         : fmt == 4u ? 0x611u : fmt == 5u ? 0x611611u : 0x1u;
    const uint32_t nsub = fmt == 1u ? 3u : fmt == 2u ? 4u : fmt == 3u ? 6u : fmt == 4u ? 3u : fmt == 5u ? 6u : 1u;
    const uint32_t color = fmt == 1u ? 0x3u : fmt == 2u ? 0x6u : fmt == 3u ? 0x18u : fmt == 4u ? 0x3u : fmt == 5u ? 0x1Bu : 0u;
    const uint32_t bb = (fmt == 1u || fmt == 4u) ? 8u : (fmt == 0u || fmt > 5u) ? 1u : 16u;
    const uint32_t px = (fmt >= 1u && fmt <= 5u) ? 4u : 1u;
    const bool aligned = ((w0 >> 1) & 1u) != 0u;
    t.format = (fmt >= 1u && fmt <= 5u) ? fmt : 0u;
    t.precon = 1; t.swizzle = w0 & 1u; t.block_bytes = bb; t.num_sub = nsub; t.color_mask = color;
    t.num_mips = ((w1 >> 8) & 0x1Fu) + 1u;
    uint32_t off = 0;
#pragma unroll
    for (uint32_t i = 0; i < kMaxSubBlocks; ++i) {
        t.sub_size[i] = i < nsub ? (sizes >> (4u * i)) & 15u : 0u;
        t.sub_off[i] = off; off += t.sub_size[i];
    }
    t.w[0] = ((w0 >> 2) & 0x7FFFu) + 1u; t.h[0] = ((w0 >> 17) & 0x7FFFu) + 1u;
    t.pitch[0] = ((w1 >> 13) & 0x7FFFFu) + 1u;
    uint32_t mw = (t.w[0] * px) / 2u, mh = (t.h[0] * px) / 2u;
    t.mip_off_bytes[0] = 0; t.mip_off_blocks[0] = 0; t.item_prefix[0] = 0;
    // All sizes are accumulated in 64 bits and must stay inside the stream's output: the header fields are
    // 15 + 15 + 19 bits wide, so pitch * h alone can pass 2^32 (the reference computes in 32 bits and would
    // wrap, inc/common/BrotligDataConditioner.h:204-219; a wrapped total that happens to equal out_size must
    // not be accepted, the de-conditioning kernel walks the real rows).
    uint64_t total = 0, bytes = 0, items = 0;
    bool fits = true;
    for (uint32_t m = 0; m < t.num_mips; ++m) {
        if (m > 0u) {                                                   // :204-210
            t.w[m] = (mw + px - 1u) / px; t.h[m] = (mh + px - 1u) / px;
            const uint32_t row = t.w[m] * bb;
            t.pitch[m] = aligned ? (row + 255u) / 256u * 256u : row;
            mw /= 2u; mh /= 2u;
        }
        if ((uint64_t)t.pitch[m] < (uint64_t)t.w[m] * bb) fits = false;
        total += (uint64_t)t.w[m] * t.h[m];
        bytes += (uint64_t)t.pitch[m] * t.h[m];
        const uint32_t rg = dc_row_group_log2(t.w[m], t.h[m], t.pitch[m], bb, t.swizzle);                       // (narrow mips: 128 / W tile rows per super-tile)
        items += (uint64_t)((((t.h[m] + 1u) / 2u) + (1u << rg) - 1u) >> rg) * ((((t.pitch[m] + bb - 1u) / bb + 31u) / 32u + 3u) / 4u) * 256u;    // tile rows of whole super-tiles (4 tiles)
        if (bytes > (uint64_t)out_size || total * bb > (uint64_t)out_size || items > 0xFFFFFFFFull) fits = false;
        t.mip_off_bytes[m + 1] = fits ? (uint32_t)bytes : 0u;
        t.mip_off_blocks[m + 1] = fits ? (uint32_t)total : 0u;
        t.item_prefix[m + 1] = fits ? (uint32_t)items : 0u;
    }
    for (uint32_t m = t.num_mips; m < kMaxMips; ++m) { t.w[m] = t.h[m] = t.pitch[m] = 0; }
    if (!fits) return false;
    t.total_blocks = (uint32_t)total; t.tex_bytes = (uint32_t)(total * bb);
    t.sub_stream_off[0] = 0;
#pragma unroll
    for (uint32_t i = 0; i < kMaxSubBlocks; ++i) t.sub_stream_off[i + 1] = t.sub_stream_off[i] + t.total_blocks * t.sub_size[i];
    return bytes == (uint64_t)out_size;                                 // :219
}

// Kernel 3 (preconditioned streams only): conditioned space -> texture space.
// The reference scatters byte by byte (PageDecoder.cpp:243-265,:406-444).  Here the work item is a SUPER-TILE of 2 texture rows x 128
// block columns (256 blocks), one per wavefront and step:
//   * wide path (round 5) -- a swizzled mip with even dimensions, the super-tile full of real blocks: under the 2x2 swizzle
//     (PageDecoder.cpp:416-436) its 256 blocks are CONSECUTIVE in every conditioned sub-stream, so each sub-stream contributes one
//     contiguous segment of 256 x sub-block-size bytes.  The wavefront reads all segments with 16-byte-per-lane loads (every load
//     instruction 1 KiB of contiguous bytes; all of them in flight together), parks them in LDS (4 KiB, the segments back to back), and
//     every lane then assembles four blocks from LDS -- one typed LDS read per sub-block -- and stores them, 64 lanes x 16 bytes = 1 KiB
//     of one texture row per store instruction.  Rounds 1-4 gathered with one 1 / 2 / 4 / 6-byte load per sub-block and lane: seven load
//     instructions for the kilobyte that now takes one, and the kernel was bound by the bytes it could keep in flight that way
//     (profiles/r04_final_kernel_trace_stats_bc3.md: 2.45 ms for 4 GiB in + 4 GiB out, 56 % of the copy rate).
//   * gather path -- everything else (no swizzle, odd dimensions, the last columns of a row, row-pitch padding, small mips, unknown
//     formats): the per-block gather of rounds 1-4, two tiles of 2 x 32 blocks at a time -- one thread owns one block-sized chunk of one
//     texture row, reads that block's sub-blocks (one typed load per sub-block) and writes the chunk with one store, or zeros for
//     row-pitch padding, which the reference leaves at the 0 of its initial memset (src/BrotligDecoder.cpp:448).
// Streams are spread over blockIdx.y, a stream's super-tiles over the wavefronts of blockIdx.x.
__device__ __forceinline__ uint64_t dc_load_sub(const uint8_t* src, uint32_t sz)
{
    uint64_t v = 0;
    switch (sz) {
    case 1: v = *src; break;
    case 2: { uint16_t t; __builtin_memcpy(&t, src, 2); v = t; break; }
    case 4: { uint32_t t; __builtin_memcpy(&t, src, 4); v = t; break; }
    case 6: { uint16_t t[3]; __builtin_memcpy(t, src, 6); v = (uint64_t)t[0] | ((uint64_t)t[1] << 16) | ((uint64_t)t[2] << 32); break; }
    case 8: __builtin_memcpy(&v, src, 8); break;
    default: for (uint32_t i = 0; i < sz; ++i) v |= (uint64_t)src[i] << (8u * i); break;
    }
    return v;
}
// the sub-blocks of one block, in their order, packed into the block's 16 (or 8) bytes
template <uint32_t kSizes, uint32_t kNumSub>
__device__ __forceinline__ void dc_pack_block(const uint64_t (&v)[kNumSub], uint64_t& lo, uint64_t& hi)
{
    uint32_t off = 0;
    lo = 0; hi = 0;
#pragma unroll
    for (uint32_t sub = 0; sub < kNumSub; ++sub) {
        const uint32_t sz = (kSizes >> (4u * sub)) & 15u;
        if (off < 8u) { lo |= v[sub] << (8u * off); if (off + sz > 8u) hi |= v[sub] >> (8u * (8u - off)); }
        else hi |= v[sub] << (8u * (off - 8u));
        off += sz;
    }
}
template <uint32_t kSizes, uint32_t kNumSub> constexpr uint32_t dc_sub_off(uint32_t sub)     // bytes of a block before sub-block `sub`
{
    uint32_t off = 0;
    for (uint32_t i = 0; i < sub && i < kNumSub; ++i) off += (kSizes >> (4u * i)) & 15u;
    return off;
}

constexpr uint32_t kDcSuperCols = 128, kDcSuperBlocks = 2u * kDcSuperCols, kDcSuperTiles = kDcSuperCols / 32u;
constexpr uint32_t kDcLdsBytes = kDcSuperBlocks * 16u;             // a super-tile of 16-byte blocks
static_assert(kDcSuperTiles == 4u, "dc_init pads every tile row to whole super-tiles of four tiles");
#ifndef BROTLIG_TUNE_DC_WIDE
#define BROTLIG_TUNE_DC_WIDE 1          // 0: every super-tile through the gather path (A/B)
#endif
#ifndef BROTLIG_TUNE_DC_ASM_UNROLL
#define BROTLIG_TUNE_DC_ASM_UNROLL 1    // blocks a lane assembles from LDS side by side (of its four per super-tile): registers against LDS latency
#endif

// Gather path: the tiles tc0 .. tc0 + kTiles - 1 (2 rows x 32 chunk columns each) of tile row `tr` of mip `m`, one chunk per lane and
// tile: ALL the loads of all tiles are issued before the first is used (the gather is bound by the bytes it has in flight).
template <uint32_t kSizes, uint32_t kNumSub, uint32_t kTiles>
__device__ __forceinline__ void dc_gather_tiles(const uint32_t (&sso)[kNumSub], const uint8_t* __restrict__ cond, uint8_t* __restrict__ mip_tex,
                                                uint32_t bb, uint32_t W, uint32_t H, uint32_t pitch, uint32_t per_row, uint32_t swizzle, uint32_t mip_block0,
                                                uint32_t tr, uint32_t tc0, uint32_t l)
{
    // (the mip's geometry arrives as wave-uniform values, read from the table once per super-tile by the caller: read here, through a
    // reference, every word was a vector load per lane)
    uint8_t* const tex = mip_tex;
    uint8_t* dst[kTiles];
    uint32_t nbytes[kTiles], gblock[kTiles];
    bool valid[kTiles], loads[kTiles];
#pragma unroll
    for (uint32_t u = 0; u < kTiles; ++u) {
        valid[u] = false; loads[u] = false; dst[u] = tex; nbytes[u] = 0; gblock[u] = 0;
        const uint32_t row = 2u * tr + ((l >> 1) & 1u), col = 32u * (tc0 + u) + 2u * (l >> 2) + (l & 1u);
        if (row >= H || col >= per_row) continue;
        valid[u] = true;
        dst[u] = tex + row * pitch + col * bb;
        nbytes[u] = min_u32(bb, pitch - col * bb);
        if (col < W) {
            // inverse of the 2x2 de-swizzle (PageDecoder.cpp:416-436): texture (row, col) -> block index
            uint32_t block = row * W + col;
            const uint32_t effW = W - (W & 1u), effH = H - (H & 1u);
            if (swizzle && W >= 2u && H >= 2u && row < effH && col < effW) {
                // eff = (row / 2) * 2 effW + x with x = 4 (col / 2) + 2 (row & 1) + (col & 1) < 2 effW,
                // so eff / effW and eff % effW need one compare, not a division
                const uint32_t x = (col >> 1) * 4u + (row & 1u) * 2u + (col & 1u);
                const uint32_t wrap = x >= effW ? 1u : 0u;
                block = (2u * (row >> 1) + wrap) * W + (x - (wrap ? effW : 0u));
            }
            gblock[u] = mip_block0 + block;
            loads[u] = true;
        }
    }
    // sub-block sizes known at compile time: every load of every tile is issued here, back to back, and waited for once
    uint64_t v[kTiles][kNumSub];
#pragma unroll
    for (uint32_t u = 0; u < kTiles; ++u) {
#pragma unroll
        for (uint32_t sub = 0; sub < kNumSub; ++sub) {
            const uint32_t sz = (kSizes >> (4u * sub)) & 15u;
            v[u][sub] = dc_load_sub(cond + sso[sub] + gblock[u] * sz, sz);     // (unconditional -- a lane without a block reads block 0 and drops
                                                                                // it: a branch per load keeps the loads from being in flight together)
        }
    }
#pragma unroll
    for (uint32_t u = 0; u < kTiles; ++u) {
        if (!valid[u]) continue;
        uint64_t lo, hi;
        dc_pack_block<kSizes, kNumSub>(v[u], lo, hi);
        if (!loads[u]) { lo = 0; hi = 0; }                                  // row-pitch padding
        uint8_t* const d = dst[u];
        const bool aligned = ((uint64_t)(uintptr_t)d & (uint64_t)(bb - 1u)) == 0u;
        if (nbytes[u] == 16u && aligned) { uint64_t q[2] = {lo, hi}; __builtin_memcpy(__builtin_assume_aligned(d, 16), q, 16); }
        else if (nbytes[u] == 8u && bb == 8u && aligned) __builtin_memcpy(__builtin_assume_aligned(d, 8), &lo, 8);
        else for (uint32_t i = 0; i < nbytes[u]; ++i) d[i] = (uint8_t)((i < 8u ? lo >> (8u * i) : hi >> (8u * (i - 8u))));
    }
}

// Super-tiles st_first, st_first + step, ... < st_end of one texture, by one wavefront; returns the first one of that progression it did not take.
// `lds`: kDcLdsBytes of this wavefront's own.
template <uint32_t kSizes, uint32_t kNumSub>
__device__ __forceinline__ uint32_t dc_texture(const DcTable* __restrict__ tp, const uint8_t* __restrict__ cond, uint8_t* __restrict__ tex,
                                               uint32_t st_first, uint32_t st_end, uint32_t step, uint8_t* lds)
{
    // (the table -- written by the schedule kernel, constant here -- is read through the constant address space: every word a scalar load.  As
    // plain global memory, even behind __restrict__, its words came as one vector load per lane each, waited for in front of the loads they
    // are the addresses of, and kept in vector registers)
    const BROTLIG_CONSTANT_AS DcTable& t = *(const BROTLIG_CONSTANT_AS DcTable*)tp;
    constexpr uint32_t bbK = dc_sub_off<kSizes, kNumSub>(kNumSub);         // block bytes of the format: 8 or 16 (1 for the unknown format)
    const uint32_t lane = wave::lane_id();
    const uint32_t bb = t.block_bytes;
    uint32_t sso[kNumSub];
#pragma unroll
    for (uint32_t sub = 0; sub < kNumSub; ++sub) sso[sub] = t.sub_stream_off[sub];
    // (item_prefix counts lanes x tiles -- 64 per tile of 2 x 32 chunks --, every tile row padded to whole super-tiles: >> 8 = super-tiles)
    uint32_t m = 0;
    uint32_t st0 = st_first;
    for (; st0 < st_end; st0 += step) {
        // the mip and super-tile coordinates are wave-uniform and go to the scalar unit
        const uint32_t st = wave::uniform(st0);
        while (m + 1u < kMaxMips && (st << 8) >= t.item_prefix[m + 1]) ++m;     // (bounded by the table whatever it holds)
        const uint32_t W = t.w[m], H = t.h[m], pitch = t.pitch[m];
        const uint32_t mip_bytes0 = t.mip_off_bytes[m], mip_block0 = t.mip_off_blocks[m], swizzle = t.swizzle;
        const uint32_t per_row = (pitch + bb - 1u) / bb, tiles_x = (per_row + 31u) / 32u, supers_x = (tiles_x + kDcSuperTiles - 1u) / kDcSuperTiles;
        const uint32_t local = st - (t.item_prefix[m] >> 8);
        const uint32_t trg = local / supers_x, q = local - trg * supers_x;
        // the super-tile's tile rows (pairs of texture rows): one, or 128 / W of a narrow mip (then supers_x is 1 and q is 0)
        const uint32_t rg = dc_row_group_log2(W, H, pitch, bb, swizzle);
        const uint32_t tile_rows = (H + 1u) >> 1, tr0 = trg << rg, tr1 = ((trg + 1u) << rg) < tile_rows ? (trg + 1u) << rg : tile_rows;
        const uint32_t lw = 7u - rg;                                            // log2 of the super-tile's columns
        const bool wide = BROTLIG_TUNE_DC_WIDE && bbK >= 8u && bb == bbK && swizzle != 0u && ((W | H) & 1u) == 0u && ((mip_bytes0 | pitch) & (bbK - 1u)) == 0u &&
                          (rg ? ((trg + 1u) << rg) <= tile_rows : (2u * tr0 + 1u < H && kDcSuperCols * (q + 1u) <= W));
        if (wide) {
            if constexpr (bbK >= 8u) {
                // first block of the super-tile in conditioned order: block(row, col) = 2 (row / 2) W + 4 (col / 2) + 2 (row & 1) + (col & 1)
                const uint32_t g0 = mip_block0 + 2u * tr0 * W + kDcSuperBlocks * q;
                constexpr uint32_t kLoads = bbK / 4u;                           // 16-byte units: 16 bbK of them, 64 per load instruction
                Bytes16 seg[kLoads];
#pragma unroll
                for (uint32_t i = 0; i < kLoads; ++i) {
                    const uint32_t u = 64u * i + lane;                          // unit u = LDS bytes [16 u, 16 u + 16): the segments back to back
                    uint32_t src = sso[0] + g0 * (kSizes & 15u) + 16u * u;
#pragma unroll
                    for (uint32_t sub = 1; sub < kNumSub; ++sub) {
                        const uint32_t first = 16u * dc_sub_off<kSizes, kNumSub>(sub);         // first unit of segment `sub`
                        const uint32_t sz = (kSizes >> (4u * sub)) & 15u;
                        if (u >= first) src = sso[sub] + g0 * sz + 16u * (u - first);
                    }
                    __builtin_memcpy(&seg[i], cond + src, 16);                  // (any byte alignment: the sub-streams start where they start)
                }
#pragma unroll
                for (uint32_t i = 0; i < kLoads; ++i) store16(lds + 16u * (64u * i + lane), seg[i]);
                wave::sync();
                uint8_t* const row0 = tex + mip_bytes0 + 2u * tr0 * pitch + kDcSuperCols * q * bbK;
                const uint32_t cmask = (1u << lw) - 1u;
#pragma unroll BROTLIG_TUNE_DC_ASM_UNROLL
                for (uint32_t i = 0; i < kDcSuperBlocks / 64u; ++i) {
                    // block idx of the super-tile in TEXTURE order: row r (of 2 .. 128), column c (of 128 .. 2)
                    const uint32_t idx = 64u * i + lane, r = idx >> lw, c = idx & cmask;
                    const uint32_t j = ((r >> 1) << (lw + 1u)) + 4u * (c >> 1) + 2u * (r & 1u) + (c & 1u);       // its place among the 256, conditioned order
                    uint64_t v[kNumSub];
#pragma unroll
                    for (uint32_t sub = 0; sub < kNumSub; ++sub) {
                        const uint32_t sz = (kSizes >> (4u * sub)) & 15u;
                        v[sub] = dc_load_sub(lds + kDcSuperBlocks * dc_sub_off<kSizes, kNumSub>(sub) + j * sz, sz);
                    }
                    uint64_t lo, hi;
                    dc_pack_block<kSizes, kNumSub>(v, lo, hi);
                    uint8_t* const d = row0 + r * pitch + c * bbK;
                    if constexpr (bbK == 16u) { uint64_t w[2] = {lo, hi}; __builtin_memcpy(__builtin_assume_aligned(d, 16), w, 16); }
                    else __builtin_memcpy(__builtin_assume_aligned(d, 8), &lo, 8);
                }
                wave::sync();                                                   // the next super-tile overwrites the segments
            }
        } else {
            // two tiles at a time (four at once cost more registers than their loads in flight bring: round 4, 79 VGPRs)
#pragma nounroll
            for (uint32_t tr = tr0; tr < tr1; ++tr) {
#pragma nounroll
                for (uint32_t tc = kDcSuperTiles * q; tc < kDcSuperTiles * (q + 1u) && tc < tiles_x; tc += 2u)
                    dc_gather_tiles<kSizes, kNumSub, 2u>(sso, cond, tex + mip_bytes0, bb, W, H, pitch, per_row, swizzle, mip_block0, tr, tc, lane);
            }
        }
    }
    return st0;
}

// log2 of the wavefronts of a gang of the de-conditioning kernel; -1: by the batch (see the kernel)
#ifndef BROTLIG_TUNE_DC_GANG_LOG2
#define BROTLIG_TUNE_DC_GANG_LOG2 -1
#endif
// Super-tiles first, first + step, ... < end of the BATCH's list (DcTable::super_base says where a stream's begin), by one wavefront.
__device__ __forceinline__ void dc_walk(const DecodeArgs& a, uint32_t first, uint32_t end, uint32_t step, uint8_t* lds)
{
    if (first >= end) return;
    // (the tables were written by the schedule kernel and are constant here: wave-uniform words through the constant address space are
    // scalar loads)
    const BROTLIG_CONSTANT_AS DcTable* const dc = (const BROTLIG_CONSTANT_AS DcTable*)a.dc;
    // the stream `first` falls into: the last one whose super-tiles begin at or before it (streams without any share their successor's base and sort before it)
    uint32_t s = 0;
    for (uint32_t hi = a.num_streams; hi - s > 1u;) { const uint32_t mid = (s + hi) >> 1; if (dc[mid].super_base <= first) s = mid; else hi = mid; }
    uint32_t i = first;
    while (i < end && s < a.num_streams) {
        const BROTLIG_CONSTANT_AS DcTable& t = dc[s];
        const uint32_t base = t.super_base;
        const uint32_t supers = t.precon ? t.item_prefix[t.num_mips] >> 8 : 0u;
        if (supers == 0u || base + supers <= i || base > i) { ++s; continue; }        // (base > i: a table that is not a prefix -- nothing is touched)
        const uint32_t st_first = i - base, st_end = end - base < supers ? end - base : supers;
        const uint64_t off = a.streams[s].out_offset;
        const uint8_t* cond = a.scratch + off;
        uint8_t* tex = a.out + off;
        const DcTable* tp = a.dc + s;
        uint32_t next;
        // per-format instantiations (sub-block sizes, four bits each, first sub-block lowest: dc_init)
        switch (t.format) {
        case 1: next = dc_texture<0x422u, 3u>(tp, cond, tex, st_first, st_end, step, lds); break;
        case 2: next = dc_texture<0x4228u, 4u>(tp, cond, tex, st_first, st_end, step, lds); break;
        case 3: next = dc_texture<0x422611u, 6u>(tp, cond, tex, st_first, st_end, step, lds); break;
        case 4: next = dc_texture<0x611u, 3u>(tp, cond, tex, st_first, st_end, step, lds); break;
        case 5: next = dc_texture<0x611611u, 6u>(tp, cond, tex, st_first, st_end, step, lds); break;
        default: next = dc_texture<0x1u, 1u>(tp, cond, tex, st_first, st_end, step, lds); break;
        }
        i = base + next;
        ++s;
    }
}

__global__ void __launch_bounds__(64) brotlig_decondition_kernel(DecodeArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t seg_lds[kDcLdsBytes];
    // The batch's super-tiles are ONE list -- stream after stream -- cut into equal runs, one per GANG of wavefronts (consecutive
    // workgroups); the gang's wavefronts take the run's super-tiles in turn, and a run begins and ends wherever it does -- inside a texture,
    // inside a mip, across streams that are not pre-conditioned.  Large textures (1 024 super-tiles = 4 MiB of BC3 and more on average) are
    // walked by gangs of 256: at any moment a gang reads and writes one neighbourhood, and HBM sees a few dozen long sequential streams
    // instead of one per wavefront (config 4: prepare + de-conditioning 1.68 ms with gangs of 256, 1.78 with gangs of one); anything smaller
    // by gangs of one, a contiguous run per wavefront (4 096 textures of 64 KiB: 0.18 ms against 0.32).  (Until late in round 5 the streams
    // were spread over blockIdx.y and each stream's super-tiles over the 256 wavefronts of blockIdx.x: right for the benchmark's 16 MiB
    // textures -- 1.63 ms --, but a 64 KiB texture has 16 super-tiles: 4 096 of them took 0.53 ms, 1.0 TB/s instead of 5, a third of that
    // batch's whole step; profiles/experiments/r05_many_textures.md.)
    const uint32_t total = wave::uniform(a.status[5]);
    if (total == 0u) return;                                            // no preconditioned stream in this batch
    const uint32_t textures = wave::uniform(a.status[2]);
    const uint32_t gl = BROTLIG_TUNE_DC_GANG_LOG2 >= 0 ? (uint32_t)BROTLIG_TUNE_DC_GANG_LOG2 : ((total >> 10) >= textures ? 8u : 0u);
    const uint32_t G = 1u << gl;
    const uint32_t gangs = (gridDim.x + G - 1u) >> gl, gang = blockIdx.x >> gl, member = blockIdx.x - (gang << gl);
    const uint32_t members = gang + 1u < gangs ? G : gridDim.x - (gang << gl);
    const uint32_t per_gang = (total + gangs - 1u) / gangs;
    const uint32_t lo = wave::uniform(gang * per_gang);
    if (lo >= total) return;
    const uint32_t hi = per_gang < total - lo ? lo + per_gang : total;
    // Gangs that run side by side start at different places of their runs: equal textures sit at power-of-two distances in memory, and
    // walking them in step would hit the same HBM channels.  The start is a whole number of turns into the run (a multiply-high, not a
    // remainder: DESIGN 6.0), the part before it comes last.
    const uint32_t turns = (hi - lo) / members;                         // (32-bit operands: the exact division)
    const uint32_t mid = lo + members * (uint32_t)(((uint64_t)(gang * 2654435761u) * turns) >> 32);
    dc_walk(a, mid + member, hi, members, seg_lds);
    dc_walk(a, lo + member, mid, members, seg_lds);
}

}  // namespace brotlig
