// brotlig_cpu.cpp -- DecodeCPU of the drop-in boundary (libbrotlig_cpu.so), SURVEY.md 8(b) row 2.
//
// Replaces: BROTLIG_ERROR BrotliG::DecodeCPU(uint32_t input_size, const uint8_t* src, uint32_t* output_size,
//                                            uint8_t* output, BROTLIG_Feedback_Proc feedbackProc)
//   inc/BrotligDecoder.h:33, src/BrotligDecoder.cpp:426-519, src/decoder/PageDecoder.cpp:65-268.
// A separate library from libbrotlig_hip.so on purpose: the GPU path has no CPU fallback and never loads this.
// It shares nothing with the repo's test checker either (that one restates the reference's cost profile -- six
// 64 KiB tables filled per page -- and stays the checker and the reported CPU baseline); this one is written for
// speed from the format: one 11-bit lookup per symbol with a canonical search behind it, 64-bit bit
// windows refilled eight bytes at a time, pages fanned out over std::thread workers on one atomic counter,
// de-conditioning as a block gather after the pages instead of a per-byte scatter.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "brotlig_amd_cpu.h"

namespace {

#define __host__
#define __device__
#include "brotlig_format.h"
#include "brotlig_shard_plan.h"
#undef __host__
#undef __device__
using namespace brotlig;

// RFC 7932 section 5 (the reference carries them as sBrotligCmdLut, inc/common/BrotligCommandLut.h:41-747)
const uint32_t kInsBase[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594};
const uint8_t kInsExtra[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
const uint32_t kCopyBase[24] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118};
const uint8_t kCopyExtra[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};
const uint8_t kCodeLenOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};   // BrotligHuffmanTable.cpp:40-42

inline uint32_t bit_width(uint32_t x) { return x ? 32u - (uint32_t)__builtin_clz(x) : 0u; }
inline uint32_t rev_bits(uint32_t v, uint32_t n)
{
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

// One sub-bitstream: LSB-first, 64-bit window, never reads outside [lo, hi) (bytes beyond read as zero, like the
// padded tail the reference over-reads into, inc/common/BrotligDeswizzler.h:74-81).
struct SubReader {
    const uint8_t* base; size_t pos, end;
    uint64_t buf; uint32_t avail;
    void init(const uint8_t* b, size_t start, size_t limit) { base = b; pos = start; end = limit; buf = 0; avail = 0; refill(); }
    void refill()
    {
        if (pos + 8 <= end) {
            uint64_t w; memcpy(&w, base + pos, 8);
            buf |= w << avail;
            const uint32_t take = (63u - avail) >> 3;
            pos += take; avail += take * 8u;
        } else {
            while (avail <= 56u) { const uint64_t b = pos < end ? base[pos] : 0u; buf |= b << avail; ++pos; avail += 8u; }
        }
    }
    uint32_t peek(uint32_t n) { if (avail < n) refill(); return (uint32_t)(buf & ((1ull << n) - 1ull)); }     // n <= 32
    void consume(uint32_t n) { buf >>= n; avail -= n; }
    uint32_t read(uint32_t n) { if (n == 0) return 0; const uint32_t v = peek(n); consume(n); return v; }
};

// Canonical prefix code: 11-bit primary table {symbol << 4 | length}, longer codes by a search over the lengths.
constexpr uint32_t kRootBits = 11;
struct Code {
    uint16_t root[1u << kRootBits];
    uint16_t sorted[kIcpAlphabet];          // symbols in (length, symbol) order
    uint32_t limit[17], first[17], offs[17]; // per length: exclusive upper bound / first code (left-justified, 15 bits), index of its first symbol
    uint32_t maxlen;
    bool single; uint32_t single_sym;       // trivial code: one symbol, zero bits

    void from_lengths(const uint8_t* len, uint32_t alphabet)
    {
        single = false;
        uint32_t count[16] = {0};
        for (uint32_t s = 0; s < alphabet; ++s) ++count[len[s]];
        count[0] = 0;
        uint32_t code = 0, off = 0, next[16];
        maxlen = 0;
        for (uint32_t l = 1; l < 16; ++l) {
            code = (code + count[l - 1]) << 1;
            first[l] = code << (15 - l); offs[l] = off; next[l] = off;
            limit[l] = (code + count[l]) << (15 - l);
            if (limit[l] > 32768u) limit[l] = 32768u;
            if (first[l] > 32768u) first[l] = 32768u;
            off += count[l];
            if (count[l]) maxlen = l;
        }
        for (uint32_t s = 0; s < alphabet; ++s) if (len[s]) sorted[next[len[s]]++] = (uint16_t)s;
        // primary table: every root index whose first kRootBits code bits start a code of length <= kRootBits
        for (uint32_t e = 0; e < (1u << kRootBits); ++e) root[e] = 0;           // 0 = "longer code or none"
        for (uint32_t l = 1; l <= kRootBits && l <= maxlen; ++l) {
            for (uint32_t k = 0; k < count[l]; ++k) {
                const uint32_t c = (first[l] >> (15 - l)) + k;                  // the code, MSB-first
                const uint16_t entry = (uint16_t)((sorted[offs[l] + k] << 4) | l);
                for (uint32_t e = rev_bits(c, l); e < (1u << kRootBits); e += 1u << l) root[e] = entry;
            }
        }
    }
    // Decodes one symbol; false when the bits match no code (corrupt stream).
    bool decode(SubReader& r, uint32_t& sym)
    {
        if (single) { sym = single_sym; return true; }
        const uint32_t bits = r.peek(15);
        const uint32_t e = root[bits & ((1u << kRootBits) - 1u)];
        if (e) { sym = e >> 4; r.consume(e & 15u); return true; }
        const uint32_t v = rev_bits(bits, 15);                                  // first stream bit on top
        for (uint32_t l = kRootBits + 1; l <= maxlen; ++l)
            if (v < limit[l]) {
                if (v < first[l]) return false;
                sym = sorted[offs[l] + ((v - first[l]) >> (15 - l))];
                r.consume(l);
                return true;
            }
        return false;
    }
};

// Forward copy in kChunk-byte pieces; source and destination may overlap when they are >= kChunk apart.  Writes up to
// kChunk - 1 bytes past d + n.
template <uint32_t kChunk>
inline void chunk_copy(uint8_t* d, const uint8_t* s, uint32_t n)
{
    for (uint32_t i = 0; i < n; i += kChunk) memcpy(d + i, s + i, kChunk);
}

struct PageCtx {
    SubReader sub[kNumStreams];
    Code icp, dist, lit;
    std::vector<uint8_t> queue;             // literal queue (PageDecoder.cpp:164-166)
};

// SURVEY.md A.5 / src/decoder/BrotligHuffmanTable.cpp:73-205: the description is read round-robin over the 32
// sub-streams starting at sub-stream 0.
bool read_code(PageCtx& P, Code& c, uint32_t alphabet)
{
    const uint32_t maxbits = bit_width(alphabet - 1);
    const uint32_t hdr = P.sub[0].read(6);
    const uint32_t type = hdr & 3u;
    if (type == 0u) {                                                           // trivial: one symbol
        c.single = true; c.single_sym = P.sub[0].read(maxbits); c.maxlen = 0;
        return c.single_sym < alphabet;
    }
    uint8_t len[kIcpAlphabet];
    memset(len, 0, sizeof len);
    if (type == 1u) {                                                           // simple: 2..4 symbols, fixed shapes (:26-38,:98-120)
        const uint32_t nsym = ((hdr >> 2) & 3u) + 1u, tree_select = (hdr >> 4) & 1u;
        // one symbol is not a simple code: the reference indexes FixedCodelengths[nsym - 2] (out of bounds,
        // BrotligHuffmanTable.cpp:103) and no encoder writes it -- rejected here and in the GPU kernel alike
        if (nsym < 2u) return false;
        // code lengths per shape: {1,1} {1,2,2} {2,2,2,2} {1,2,3,3}
        const uint32_t shape = nsym < 4u ? nsym - 2u : (tree_select ? 3u : 2u);
        uint32_t syms[4];
        for (uint32_t k = 0; k < nsym; ++k) { syms[k] = P.sub[k].read(maxbits); if (syms[k] >= alphabet) return false; }
        // the k-th symbol read gets the k-th code of the shape: codes are assigned in reading order, not symbol order
        c.single = false;
        for (uint32_t e = 0; e < (1u << kRootBits); ++e) {
            const uint32_t b0 = e & 1u, b1 = (e >> 1) & 1u, b2 = (e >> 2) & 1u;
            uint32_t k, l;
            if (shape == 0u) { k = b0; l = 1; }
            else if (shape == 1u) { k = b0 ? 1u + b1 : 0u; l = b0 ? 2u : 1u; }
            else if (shape == 2u) { k = b0 * 2u + b1; l = 2; }
            else { k = !b0 ? 0u : (!b1 ? 1u : 2u + b2); l = !b0 ? 1u : (!b1 ? 2u : 3u); }
            c.root[e] = (uint16_t)((syms[k < nsym ? k : 0] << 4) | l);
        }
        c.maxlen = 3;
        for (uint32_t l = 0; l < 17; ++l) { c.limit[l] = 0; c.first[l] = 0; c.offs[l] = 0; }
        return true;
    }
    // complex: code-length code, then run-length coded code lengths (:121-199)
    const uint32_t ncl = ((hdr >> 2) & 15u) + 4u > 18u ? 18u : ((hdr >> 2) & 15u) + 4u;
    uint8_t cl_len[18];
    memset(cl_len, 0, sizeof cl_len);
    for (uint32_t k = 0; k < ncl; ++k) {
        uint32_t l = P.sub[k].read(5);
        if (l > 9u) l = 0u;                                                     // the reference's table has 2^9 entries
        cl_len[kCodeLenOrder[k]] = (uint8_t)l;
    }
    // 9-bit table for the code-length code (canonical in (length, symbol) order)
    uint16_t cl_tab[512];
    memset(cl_tab, 0, sizeof cl_tab);
    {
        uint32_t code = 0;
        for (uint32_t l = 1; l <= 9; ++l) {
            for (uint32_t s = 0; s < 18; ++s) if (cl_len[s] == l) {
                for (uint32_t e = rev_bits(code, l); e < 512u; e += 1u << l) cl_tab[e] = (uint16_t)((s << 4) | l);
                ++code;
            }
            code <<= 1;
        }
    }
    uint32_t produced = 0, prev = 8;                                            // BROTLI_INITIAL_REPEATED_CODE_LENGTH
    for (uint32_t j = 0; produced < alphabet; ++j) {
        SubReader& r = P.sub[j & 31u];
        const uint32_t e = cl_tab[r.peek(9)];
        const uint32_t s = e >> 4, l = e & 15u;
        if (l == 0u) return false;                                              // no such code: corrupt
        r.consume(l);
        uint32_t run = 1, value = s;
        if (s == 16u) { run = 3u + r.read(2); value = prev; }
        else if (s == 17u) { run = 3u + r.read(3); value = 0; }
        else prev = s;
        for (uint32_t k = 0; k < run && produced < alphabet; ++k) len[produced++] = (uint8_t)value;
    }
    c.from_lengths(len, alphabet);
    return true;
}

// Pre-conditioning geometry (inc/common/BrotligDataConditioner.h:92-237), all sizes in 64 bits.
struct Dc {
    bool on = false, swizzle = false;
    uint32_t block_bytes = 1, num_sub = 1, num_mips = 1, color_mask = 0;
    uint32_t sub_size[kMaxSubBlocks] = {1, 0, 0, 0, 0, 0}, sub_off[kMaxSubBlocks] = {0};
    uint64_t sub_stream_off[kMaxSubBlocks + 1] = {0};
    uint32_t w[kMaxMips], h[kMaxMips], pitch[kMaxMips];
    uint64_t mip_bytes[kMaxMips + 1], mip_blocks[kMaxMips + 1];
    bool init(uint32_t w0, uint32_t w1, uint64_t out_size)
    {
        const uint32_t fmt = w1 & 0xFFu;
        static const uint8_t sizes[6][6] = {{1, 0, 0, 0, 0, 0}, {2, 2, 4, 0, 0, 0}, {8, 2, 2, 4, 0, 0}, {1, 1, 6, 2, 2, 4}, {1, 1, 6, 0, 0, 0}, {1, 1, 6, 1, 1, 6}};
        static const uint8_t nsub[6] = {1, 3, 4, 6, 3, 6}, colors[6] = {0, 0x3, 0x6, 0x18, 0x3, 0x1B}, bytes[6] = {1, 8, 16, 16, 8, 16};
        const uint32_t f = fmt <= 5u ? fmt : 0u, px = f ? 4u : 1u;
        on = true; swizzle = (w0 & 1u) != 0u;
        const bool aligned = ((w0 >> 1) & 1u) != 0u;
        block_bytes = bytes[f]; num_sub = nsub[f]; color_mask = colors[f];
        uint32_t off = 0;
        for (uint32_t i = 0; i < kMaxSubBlocks; ++i) { sub_size[i] = sizes[f][i]; sub_off[i] = off; off += sub_size[i]; }
        num_mips = ((w1 >> 8) & 0x1Fu) + 1u;
        w[0] = ((w0 >> 2) & 0x7FFFu) + 1u; h[0] = ((w0 >> 17) & 0x7FFFu) + 1u; pitch[0] = ((w1 >> 13) & 0x7FFFFu) + 1u;
        uint32_t mw = (w[0] * px) / 2u, mh = (h[0] * px) / 2u;
        uint64_t blocks = 0, total = 0;
        mip_bytes[0] = 0; mip_blocks[0] = 0;
        for (uint32_t m = 0; m < num_mips; ++m) {
            if (m) {
                w[m] = (mw + px - 1u) / px; h[m] = (mh + px - 1u) / px;
                const uint32_t row = w[m] * block_bytes;
                pitch[m] = aligned ? (row + 255u) / 256u * 256u : row;
                mw /= 2u; mh /= 2u;
            }
            if ((uint64_t)pitch[m] < (uint64_t)w[m] * block_bytes) return false;
            blocks += (uint64_t)w[m] * h[m]; total += (uint64_t)pitch[m] * h[m];
            if (total > out_size) return false;
            mip_bytes[m + 1] = total; mip_blocks[m + 1] = blocks;
        }
        sub_stream_off[0] = 0;
        for (uint32_t i = 0; i < kMaxSubBlocks; ++i) sub_stream_off[i + 1] = sub_stream_off[i] + blocks * sub_size[i];
        return total == out_size && blocks * block_bytes <= out_size;            // :219
    }
};

struct Job {
    const uint8_t* src; uint64_t src_size;
    const uint8_t* table; const uint8_t* pages; uint64_t pages_size;
    uint8_t* out; StreamInfo si; const Dc* dc;
    std::atomic<uint32_t> next{0}; std::atomic<int> error{0};
    BrotligFeedbackProc feedback = nullptr; void* user = nullptr;           // per-page progress callback (may be null)
    std::atomic<int> aborted{0};
    uint8_t* done = nullptr;                                                // [num_pages] set by the worker that completed the page
};

inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

// One page (src/decoder/PageDecoder.cpp:65-268).  `dst` receives out_size bytes (conditioned space when dc->on).
bool decode_page(PageCtx& P, const uint8_t* in, uint32_t in_size, uint64_t in_room, uint8_t* dst, uint32_t out_size,
                 uint32_t page_size, uint32_t page_off, const Dc* dc)
{
    if (in_size > in_room) return false;
    if (in_size == out_size) { memcpy(dst, in, out_size); return true; }        // stored page (:70-76)
    if (in_size < 4u) return false;
    // page header + sub-stream size table (:79-121)
    SubReader hr; hr.init(in, 0, in_size);
    const uint32_t npostfix = hr.read(2), ndirect = hr.read(4) << npostfix;
    const bool is_delta = hr.read(1) != 0u && dc->on;
    hr.read(1);
    const uint32_t base_bits = bit_width((in_size + 31u) / 32u), dsize_bits = bit_width(bit_width(in_size - 1u));
    const uint32_t base_size = hr.read(base_bits), delta_bits = hr.read(dsize_bits);
    size_t at = ((8u + base_bits + dsize_bits + 32u * delta_bits + 31u) / 32u) * 4u;
    for (uint32_t i = 0; i < kNumStreams; ++i) {
        const uint32_t len = base_size + hr.read(delta_bits);
        P.sub[i].init(in, at < in_size ? at : in_size, in_size);
        at += len;
    }
    if (!read_code(P, P.icp, kIcpAlphabet) || !read_code(P, P.dist, kDistAlphabet) || !read_code(P, P.lit, kLitAlphabet)) return false;

    uint32_t ring[4] = {4, 11, 15, 16};                                         // :150-153
    P.queue.resize((size_t)page_size + 64u);                                    // + slack for chunked reads
    uint8_t* const q = P.queue.data();
    size_t q_front = 0, q_back = 0;
    uint32_t pos = 0, prev_tail = 0;
    struct Cmd { uint32_t ins, copy, dist; } cmd[kNumStreams];
    for (uint32_t rounds = page_size / 64u + 4u; rounds; --rounds) {            // :174-236; every full round emits >= 64 bytes
        uint32_t n = 0, litcount = 0;
        bool sentinel = false;
        for (; n < kNumStreams; ++n) {
            SubReader& r = P.sub[n];
            uint32_t sym;
            if (!P.icp.decode(r, sym) || sym >= kIcpAlphabet) return false;
            if (sym == kSentinel) { sentinel = true; break; }
            Cmd c{0, 0, 0};
            if (sym < kSentinel) {
                const uint32_t cell = sym >> 6;
                const uint32_t ic = ((0x298500u >> (2u * cell)) & 3u) * 8u + ((sym >> 3) & 7u);
                const uint32_t cc = ((0x262444u >> (2u * cell)) & 3u) * 8u + (sym & 7u);
                c.ins = kInsBase[ic] + r.read(kInsExtra[ic]);
                c.copy = kCopyBase[cc] + r.read(kCopyExtra[cc]);
                uint32_t dcode = 0;
                if (sym >= 128u && (!P.dist.decode(r, dcode) || dcode >= kDistAlphabet)) return false;
                if (dcode < 16u) {                                              // :345-364
                    static const int8_t idx[16] = {0, 1, 2, 3, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1};
                    static const int8_t add[16] = {0, 0, 0, 0, -1, 1, -2, 2, -3, 3, -1, 1, -2, 2, -3, 3};
                    c.dist = ring[idx[dcode]] + (uint32_t)(int32_t)add[dcode];
                } else if (dcode < 16u + ndirect) {
                    c.dist = dcode - 15u;
                } else {                                                        // :365-390
                    const uint32_t x = dcode - ndirect - 16u;
                    uint32_t nbits = 1u + (x >> (npostfix + 1u));
                    if (nbits > 24u) nbits = 24u;
                    const uint32_t extra = r.read(nbits);
                    const uint32_t hcode = x >> npostfix, lcode = x & ((1u << npostfix) - 1u);
                    c.dist = ((((2u + (hcode & 1u)) << nbits) - 4u + extra) << npostfix) + lcode + ndirect + 1u;
                }
                if (dcode != 0u) { ring[3] = ring[2]; ring[2] = ring[1]; ring[1] = ring[0]; ring[0] = c.dist; }
            } else {                                                            // insert-only (:308-317)
                const uint32_t ic = sym - kSentinel > 23u ? 23u : sym - kSentinel;
                c.ins = kInsBase[ic] + r.read(kInsExtra[ic]);
            }
            litcount += c.ins;
            cmd[n] = c;
        }
        // literal bookkeeping (:196-199), then the round's literals round-robin over the sub-streams (:202-206)
        const uint32_t ac = litcount > prev_tail ? litcount - prev_tail : 0u;
        const uint32_t mult = n ? (ac + n - 1u) / n : 0u;
        const uint32_t rlit = n * mult;
        if (q_back + rlit > (size_t)page_size + 32u) return false;              // a valid page never queues more (tail <= 31)
        for (uint32_t j = 0; j < rlit; ++j) {
            uint32_t sym;
            if (!P.lit.decode(P.sub[j & 31u], sym)) return false;
            q[q_back++] = (uint8_t)sym;
        }
        prev_tail = rlit + prev_tail - litcount;
        for (uint32_t k = 0; k < n; ++k) {                                      // :209-233
            const Cmd& c = cmd[k];
            if (c.ins > q_back - q_front || c.ins > out_size - pos) return false;
            const bool roomy = (uint64_t)pos + c.ins + c.copy + 16u <= out_size;   // chunked copies may overshoot by < 16 bytes: never past the page
            if (roomy) chunk_copy<16>(dst + pos, q + q_front, c.ins); else memcpy(dst + pos, q + q_front, c.ins);
            pos += c.ins; q_front += c.ins;
            if (c.copy) {
                if (c.copy > out_size - pos || c.dist == 0u || c.dist > pos) return false;
                const uint8_t* s = dst + pos - c.dist;
                if (roomy && c.dist >= 16u) chunk_copy<16>(dst + pos, s, c.copy);
                else if (roomy && c.dist >= 8u) chunk_copy<8>(dst + pos, s, c.copy);
                else if (c.dist >= c.copy) memcpy(dst + pos, s, c.copy);
                else for (uint32_t j = 0; j < c.copy; ++j) dst[pos + j] = s[j];  // LZ77 overlap
                pos += c.copy;
            }
        }
        if (q_front == q_back) q_front = q_back = 0;
        else if (q_front > page_size / 2u) { memmove(q, q + q_front, q_back - q_front); q_back -= q_front; q_front = 0; }
        if (sentinel) break;
        if (rounds == 1u) return false;
    }
    if (pos != out_size) return false;
    if (is_delta) {                                                             // :446-471
        for (uint32_t cidx = 0; cidx < kMaxSubBlocks; ++cidx) {
            if (!((dc->color_mask >> cidx) & 1u)) continue;
            const uint64_t cs = dc->sub_stream_off[cidx], ce = dc->sub_stream_off[cidx + 1];
            const uint64_t ps = page_off, pe = (uint64_t)page_off + out_size;
            if (cs >= pe || ps >= ce) continue;
            const uint32_t lo = (uint32_t)((cs > ps ? cs : ps) - ps), hi = (uint32_t)((ce < pe ? ce : pe) - ps);
            uint8_t acc = 0;
            for (uint32_t i = lo; i < hi; ++i) { acc = (uint8_t)(acc + dst[i]); dst[i] = acc; }
        }
    }
    return true;
}

void worker(Job* J, uint8_t* cond)
{
    PageCtx P;
    const StreamInfo& si = J->si;
    for (;;) {
        const uint32_t i = J->next.fetch_add(1);
        if (i >= si.num_pages || J->error.load() || J->aborted.load()) return;
        const uint64_t off = i ? rd32(J->table + 4u * i) : 0u;                  // src/BrotligDecoder.cpp:310
        const uint64_t size = i + 1u < si.num_pages ? (uint64_t)rd32(J->table + 4u * (i + 1u)) - off : rd32(J->table);
        const uint32_t out_size = (i + 1u == si.num_pages && si.last_page_size) ? si.last_page_size : si.page_size;
        uint8_t* dst = (J->dc->on ? cond : J->out) + (uint64_t)i * si.page_size;
        if (off > J->pages_size || size > 0xFFFFFFFFull ||
            !decode_page(P, J->pages + off, (uint32_t)size, J->pages_size - off, dst, out_size, si.page_size, i * si.page_size, J->dc))
            J->error.store(1);
        else J->done[i] = 1;
        if (J->feedback) {                                                      // src/BrotligDecoder.cpp:318-325
            char msg[48];
            snprintf(msg, sizeof msg, "%f", 100.f * ((float)i / (float)si.num_pages));     // std::to_string(float)
            if (J->feedback(BROTLIG_PROGRESS, msg, J->user)) { J->aborted.store(1); return; }
        }
    }
}
// what a pool thread runs: nothing may leave a thread function (it would terminate the host process) -- a bad_alloc of the
// page context or of its literal queue becomes the job's error
void pool_worker(Job* J, uint8_t* cond)
{
    try { worker(J, cond); }
    catch (...) { J->error.store(1); }
}

// conditioned space -> texture, one block at a time (the inverse of PageDecoder.cpp:406-444's address map)
void decondition(const Dc& dc, const uint8_t* cond, uint8_t* tex)
{
    for (uint32_t m = 0; m < dc.num_mips; ++m) {
        const uint32_t W = dc.w[m], H = dc.h[m];
        const uint32_t effW = W - (W & 1u), effH = H - (H & 1u);
        const bool swz = dc.swizzle && W >= 2u && H >= 2u;
        for (uint32_t row = 0; row < H; ++row)
            for (uint32_t col = 0; col < W; ++col) {
                uint64_t block = (uint64_t)row * W + col;
                if (swz && row < effH && col < effW) {
                    const uint32_t x = (col >> 1) * 4u + (row & 1u) * 2u + (col & 1u);
                    const uint32_t wrap = x >= effW ? 1u : 0u;
                    block = (uint64_t)(2u * (row >> 1) + wrap) * W + (x - (wrap ? effW : 0u));
                }
                const uint64_t g = dc.mip_blocks[m] + block;
                uint8_t* d = tex + dc.mip_bytes[m] + (uint64_t)row * dc.pitch[m] + (uint64_t)col * dc.block_bytes;
                for (uint32_t s = 0; s < dc.num_sub; ++s)
                    memcpy(d + dc.sub_off[s], cond + dc.sub_stream_off[s] + g * dc.sub_size[s], dc.sub_size[s]);
            }
    }
}

}  // namespace

namespace {

BROTLIG_ERROR decode_stream(uint32_t input_size, const uint8_t* src, uint32_t* output_size, uint8_t* output, uint32_t workers,
                            BrotligFeedbackProc feedback, void* user)
{
    if (!src || !output || !output_size || input_size < 8u) return BROTLIG_ERROR_CORRUPT_STREAM;
    const uint32_t w0 = rd32(src), w1 = rd32(src + 4);
    if ((w0 & 0xFFu) != (((w0 >> 8) & 0xFFu) ^ 0xFFu)) return BROTLIG_ERROR_CORRUPT_STREAM;       // src/BrotligDecoder.cpp:437-441
    StreamInfo si;
    if (!parse_stream_header(w0, w1, si)) return BROTLIG_ERROR_INCORRECT_STREAM_FORMAT;           // :442-446
    if (si.last_page_size > si.page_size) return BROTLIG_ERROR_CORRUPT_STREAM;
    const uint64_t usize = (uint64_t)si.num_pages * si.page_size - (si.last_page_size ? si.page_size - si.last_page_size : 0u);
    if (usize > *output_size) return BROTLIG_ERROR_GENERIC;
    if ((uint64_t)si.header_bytes + 4ull * si.num_pages > input_size) return BROTLIG_ERROR_CORRUPT_STREAM;
    Dc dc;
    std::vector<uint8_t> cond;
    if (si.preconditioned) {                                                    // :466-481: the texture is *output_size bytes
        if (!dc.init(rd32(src + 8), rd32(src + 12), *output_size) || usize != *output_size) return BROTLIG_ERROR_GENERIC;
        cond.resize((size_t)si.num_pages * si.page_size);
    }
    // :448 zeroes all *output_size bytes before anything is decoded.  Same result without the extra pass over the output
    // (single-threaded, it cost a third of the 32-worker rate: 15.2 -> 10.8 GB/s, profiles/r02_/r03_final_cpu_decode.json):
    // a plain stream's pages overwrite [0, usize) entirely, so only the tail is zeroed here and, should the decode fail or
    // be aborted, the pages that were not completed (below); a texture is zeroed as a whole (row-pitch padding stays 0).
    if (si.preconditioned) memset(output, 0, *output_size);
    else memset(output + usize, 0, *output_size - usize);
    std::vector<uint8_t> done(si.num_pages, 0);
    Job J;
    J.done = done.data();
    J.src = src; J.src_size = input_size; J.table = src + si.header_bytes; J.pages = J.table + 4ull * si.num_pages;
    J.pages_size = input_size - si.header_bytes - 4ull * si.num_pages;
    J.out = output; J.si = si; J.dc = &dc; J.feedback = feedback; J.user = user;
    // default: one worker per hardware thread up to 32 -- threads are created per call; on the 256-thread host of the
    // MI355X box 32 workers decoded 15.2 GB/s and 64 and more fell back to 10 (profiles/r02_cpu_decode.json; with round 3's
    // full up-front memset: 10.8 / 14.4 with 32 / 64, profiles/r03_final_cpu_decode.json)
    uint32_t nw = workers ? workers : std::min(std::thread::hardware_concurrency(), 32u);
    if (nw == 0u) nw = 1u;
    if (nw > si.num_pages) nw = si.num_pages ? si.num_pages : 1u;
    if (nw > 128u) nw = 128u;                                                   // inc/common/BrotligConstants.h:90
    std::vector<std::thread> pool;
    pool.reserve(nw);
    // a thread that cannot be created (EAGAIN) is not an error: the workers that did start, and this thread, take
    // the pages from the shared counter
    for (uint32_t t = 1; t < nw; ++t) {
        try { pool.emplace_back(pool_worker, &J, cond.data()); }
        catch (const std::system_error&) { break; }
    }
    bool threw = false;
    try { worker(&J, cond.data()); }
    catch (...) { J.error.store(1); threw = true; }                             // (bad_alloc of the page context)
    for (auto& t : pool) t.join();
    if (threw || J.error.load() || J.aborted.load()) {
        // what the reference's up-front memset leaves of a page that was never (or not successfully) decoded: zeros
        if (!si.preconditioned)
            for (uint32_t i = 0; i < si.num_pages; ++i)
                if (!done[i]) {
                    const uint64_t o = (uint64_t)i * si.page_size;
                    memset(output + o, 0, (size_t)std::min<uint64_t>(si.page_size, usize - o));
                }
        return (threw || J.error.load()) ? BROTLIG_ERROR_GENERIC : BROTLIG_ABORTED;
    }
    if (si.preconditioned) decondition(dc, cond.data(), output);
    *output_size = (uint32_t)usize;                                             // :490
    return BROTLIG_OK;
}

// nothing C++ may leave an extern "C" entry: allocation failures become BROTLIG_ERROR_GENERIC
template <class F> BROTLIG_ERROR guarded(F&& f)
{
    try { return f(); }
    catch (...) { return BROTLIG_ERROR_GENERIC; }
}

}  // namespace

extern "C" BROTLIG_ERROR BrotligDecodeCPU(uint32_t input_size, const uint8_t* src, uint32_t* output_size, uint8_t* output, uint32_t workers)
{
    return guarded([&] { return decode_stream(input_size, src, output_size, output, workers, nullptr, nullptr); });
}

// C-safe twin of the reference's feedback path (src/BrotligDecoder.cpp:318-325): `feedback` is called once per decoded
// page, from the worker thread that decoded it, with BROTLIG_PROGRESS and the percentage as text; a non-zero return
// stops the decode (BROTLIG_ABORTED, the output is then incomplete).
extern "C" BROTLIG_ERROR BrotligDecodeCPUWithFeedback(uint32_t input_size, const uint8_t* src, uint32_t* output_size, uint8_t* output,
                                                      uint32_t workers, BrotligFeedbackProc feedback, void* user)
{
    return guarded([&] { return decode_stream(input_size, src, output_size, output, workers, feedback, user); });
}

// The multi-device shard plan (include/brotlig_amd.h), also exported here: it is host arithmetic, and a host without ROCm
// (a CPU-only rank, a scheduler) must be able to compute the same cut as the ranks that decode.
extern "C" BROTLIG_ERROR BrotligShardPlan(const uint64_t* in_sizes, uint32_t num_streams, uint32_t num_shards, uint32_t* first)
{
    return guarded([&] { return brotlig::shard_plan(in_sizes, num_streams, num_shards, first) ? BROTLIG_OK : BROTLIG_ERROR_GENERIC; });
}

// inc/BrotligDecoder.h:32, src/BrotligDecoder.cpp:35-39: no validation.  (libbrotlig_hip.so exports the same
// function; this copy makes the CPU library usable on its own, like the reference's decoder library.)
extern "C" uint32_t DecompressedSize(uint8_t* src)
{
    StreamInfo si;
    parse_stream_header(rd32(src), rd32(src + 4), si);
    return uncompressed_size(si);
}

// The reference's prototype (inc/BrotligDecoder.h:33).  Its callback type takes a std::string by value and cannot
// cross a C boundary, so this entry only accepts NULL there; callers that want progress / abort use
// BrotligDecodeCPUWithFeedback.  A non-NULL pointer is rejected rather than called with the wrong convention.
extern "C" BROTLIG_ERROR DecodeCPU(uint32_t input_size, const uint8_t* src, uint32_t* output_size, uint8_t* output, void* feedbackProc)
{
    if (feedbackProc != nullptr) return BROTLIG_ERROR_GENERIC;
    return guarded([&] { return decode_stream(input_size, src, output_size, output, 0u, nullptr, nullptr); });
}
