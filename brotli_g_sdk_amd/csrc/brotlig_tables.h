// brotlig_tables.h -- prefix-code description -> decode tables, and the symbol decode that reads them (src/decoder/BrotligHuffmanTable.cpp:73-205).
// Part of the gfx950 Brotli-G decode kernels; brotlig_kernels.h includes the parts in order and says what the whole replaces.
#pragma once
#include "brotlig_kernel_common.h"

// 1: the two fifteen-step loops of the canonical build (counters to zero; counts -> first codes and offsets) unrolled by pragma -- the product
// is built with -fno-unroll-loops, which leaves only such loops unrolled (mixed +0.45 %, text +0.3, samples16 +0.5, config 2 +0.8, 1 024 mixed
// pages one per wavefront +1.1 %; records, real files and BC3 even)
#ifndef BROTLIG_TUNE_TABLE_UNROLL
#define BROTLIG_TUNE_TABLE_UNROLL 1
#endif

namespace brotlig {

// Lanes (0..17) of a half whose code-length symbol (kCodeLenOrder[lane]) is smaller than lane sl's: the ties of the canonical order.
// kClSmaller[k] = sum over j of (kCodeLenOrder[j] < kCodeLenOrder[k]) << j; checked against the order at compile time below.  (Written out
// and selected by a chain of compares: an indexed read would be a global load per lane, in front of the table build's first LDS access.)
constexpr uint32_t kClSmaller[18] = {0x00010u, 0x00011u, 0x00013u, 0x00017u, 0x00000u, 0x0001Fu, 0x3FFBFu, 0x0003Fu, 0x3FEBFu, 0x000BFu, 0x002BFu, 0x006BFu, 0x00EBFu, 0x01EBFu, 0x03EBFu, 0x07EBFu, 0x0FEBFu, 0x1FEBFu};
constexpr bool cl_smaller_matches_the_order()
{
    constexpr uint8_t order[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};      // = kCodeLenOrder
    for (int k = 0; k < 18; ++k) {
        uint32_t mk = 0;
        for (int j = 0; j < 18; ++j) mk |= (order[j] < order[k] ? 1u : 0u) << j;
        if (mk != kClSmaller[k]) return false;
    }
    return true;
}
static_assert(cl_smaller_matches_the_order(), "kClSmaller must follow kCodeLenOrder");
template <uint32_t K> __device__ __forceinline__ uint32_t cl_smaller_select(uint32_t sl, uint32_t m)
{
    if constexpr (K < 18u) return cl_smaller_select<K + 1u>(sl, sl == K ? kClSmaller[K] : m);
    else return m;
}
__device__ __forceinline__ uint32_t cl_smaller_lanes(uint32_t sl) { return cl_smaller_select<0u>(sl, 0u); }

// One prefix-code table: which LDS arrays it lives in.
struct TableRef {
    uint16_t* lut; uint32_t* sorted; uint16_t* limit; uint32_t* first_offs;
    uint32_t alphabet; int lut_bits;
    uint16_t* far_syms;         // global overflow of `sorted` (ICP and distance tables): slot of the workgroup's first half; the
                                // second half's follows (kFarSymStride)
};
// symbols of canonical rank >= cap live in global memory, at far_syms[far_slot(alphabet) + rank - cap]
__device__ __forceinline__ uint32_t sym_cap(uint32_t alphabet)
{
    return alphabet == kIcpAlphabet ? kIcpSymCap : alphabet == kDistAlphabet ? kDistSymCap : kLitAlphabet;
}
// Offset of this half's slot behind TableRef::far_syms.  Formed where it is used (a few instructions) rather than
// carried in a register through the whole kernel: the reads are rare.
__device__ __forceinline__ uint32_t far_slot(uint32_t alphabet)
{
    return (wave::lane_id_fresh() >> 5) * kFarSymStride + (alphabet == kDistAlphabet ? kFarIcp : 0u);
}

// sorted-symbol arrays: element i lives in bits [10 * (i % 3), +10) of word i / 3
__device__ __forceinline__ uint32_t sorted_get(const uint32_t* words, uint32_t i)
{
    const uint32_t w = (i * 43691u) >> 17;                      // i / 3 for i < 98304
    return (words[w] >> (10u * (i - 3u * w))) & 0x3FFu;
}
__device__ __forceinline__ void sorted_put(uint32_t* words, uint32_t i, uint32_t sym)    // words pre-zeroed
{
    const uint32_t w = (i * 43691u) >> 17;
    atomicOr(&words[w], sym << (10u * (i - 3u * w)));
}
// the literal table (256 symbols) keeps its symbols as bytes instead
__device__ __forceinline__ uint32_t table_sym(const TableRef& t, uint32_t i)
{
    if (t.alphabet == kLitAlphabet) return (uint32_t)reinterpret_cast<const uint8_t*>(t.sorted)[i];
    const uint32_t cap = sym_cap(t.alphabet);
    if (i < cap) return sorted_get(t.sorted, i);
    // a rank the page's code never assigned (incomplete or damaged code) reads whatever an earlier page left in the slot:
    // held inside the alphabet, so that what a damaged page decodes to does not depend on the workspace's history
    return min_u32((uint32_t)t.far_syms[far_slot(t.alphabet) + (i - cap)] & 0x3FFu, t.alphabet - 1u);
}
__device__ __forceinline__ void table_set_sym(const TableRef& t, uint32_t i, uint32_t sym)   // packed words pre-zeroed
{
    if (t.alphabet == kLitAlphabet) { reinterpret_cast<uint8_t*>(t.sorted)[i] = (uint8_t)sym; return; }
    const uint32_t cap = sym_cap(t.alphabet);
    if (i < cap) sorted_put(t.sorted, i, sym);
    else t.far_syms[far_slot(t.alphabet) + (i - cap)] = (uint16_t)sym;
}

// Decode one symbol from `br` (needs avail >= 15 on entry).  Returns symbol, sets len.
// kBits = index width of the table's primary LUT.  Codes longer than that take the canonical route:
// the limits of lengths 8..15 arrive in one aligned 16-byte LDS read (same address for the whole
// half), the length is a count of compares, then one read for {first code, offset} and one for the
// symbol -- two dependent reads instead of a search loop.
template <int kBits, class Reader>
__device__ __forceinline__ uint32_t decode_symbol(const TableRef& t, const Reader& br, uint32_t& len)
{
    static_assert(kBits >= 7 && kBits <= 14, "limit words 8..15 must cover every long length");
    const uint32_t bits = (uint32_t)br.buf;
    const uint32_t e = t.lut[bits & ((1u << kBits) - 1u)];
    if (e < kLutSubtree) { len = e & 15u; return e >> 4; }
    const uint32_t rb = __brev(bits);                           // stream bits, first bit on top
    if (e != kLongCode) {                                       // all codes under this prefix share one length
        const uint32_t l = e & 15u;
        const uint32_t idx = ((e >> 4) & 0x3FFu) + ((rb >> (32u - l)) & ((1u << (l - (uint32_t)kBits)) - 1u));
        len = l;
        return table_sym(t, idx);
    }
    const uint32_t v = rb >> 17;                                // next 15 bits, MSB-first
    uint32_t lim[4];
    __builtin_memcpy(lim, t.limit + 8, 16);                     // limits of lengths 8..15, two per word
    uint32_t l = (uint32_t)kBits + 1u;
#pragma unroll
    for (int k = kBits + 1; k < 15; ++k) {
        const uint32_t w = lim[(k - 8) >> 1];
        const uint32_t lk = (k & 1) ? (w >> 16) : (w & 0xFFFFu);
        l += v >= lk ? 1u : 0u;
    }
    const uint32_t fo = t.first_offs[l];
    uint32_t idx = (fo >> 16) + ((v - (fo & 0xFFFFu)) >> (15u - l));
    idx = min_u32(idx, t.alphabet - 1u);
    len = l;
    return table_sym(t, idx);
}

// -------------------------------------------------------------------------------------------
// Prefix-code description -> decode tables (format: SURVEY.md A.5; reference reader:
// src/decoder/BrotligHuffmanTable.cpp:73-205).  Runs for both halves at once; `live` says
// whether this half has a compressed page; `codelens` = alphabet bytes of LDS for the code lengths.  Returns false for a description the format does not define (the page
// is then rejected): a `simple` code of one symbol, for which the reference indexes FixedCodelengths[-1]
// (BrotligHuffmanTable.cpp:103); DecodeCPU (csrc/brotlig_cpu.cpp) rejects the same.
template <class Reader>
__device__ inline bool build_table(const TableRef& t, uint8_t* codelens, Reader& br, bool live, uint32_t sl)
{
    const uint32_t A = t.alphabet;
    const uint32_t maxbits = bit_width_u32(A - 1u);
    const uint32_t lut_size = 1u << t.lut_bits;
    uint16_t* scratch16 = t.lut;            // LUT area doubles as scratch until the LUT itself is written

    // -- header: lane 0 of the half reads 6 bits from sub-stream 0
    uint32_t hdr = 0;
    if (live && sl == 0u) hdr = br.read(6);
    hdr = wave::half_bcast(hdr, 0);
    const uint32_t type = hdr & 3u;
    const bool is_trivial = live && type == 0u;
    const bool is_simple = live && type == 1u;
    const bool is_complex = live && type >= 2u;    // type 3 is invalid; treated as complex, fails bounds later

    // -- trivial / simple: up to 4 symbols, symbol k from sub-stream k
    const uint32_t nsym = is_trivial ? 1u : ((hdr >> 2) & 3u) + 1u;
    const bool defined = !(is_simple && nsym < 2u);
    const uint32_t tree_select = (hdr >> 4) & 1u;
    uint32_t mysym = 0;
    if ((is_trivial || is_simple) && sl < nsym) mysym = br.read(maxbits);
    const uint32_t s0 = wave::half_bcast(mysym, 0), s1 = wave::half_bcast(mysym, 1);
    const uint32_t s2 = wave::half_bcast(mysym, 2), s3 = wave::half_bcast(mysym, 3);
    // -- complex: code-length code, then RLE-coded code lengths
    if (wave::any(is_complex)) {
        // 18 code-length-code lengths, the k-th from sub-stream k, for symbols in a fixed order.  Fewer than 18
        // (header field < 14) is undefined in the reference: it builds the code-length table over the first ncl
        // symbol INDICES of an array whose other entries were never written (uninitialised stack,
        // BrotligHuffmanTable.cpp:125,:141), and its encoder always writes 18 (src/encoder/BrotligHuffman.cpp:358).
        // Here every length that was read gets its code.
        const uint32_t ncl = min_u32(((hdr >> 2) & 15u) + 4u, 18u);
        uint32_t cl_len = 0;
        const uint32_t cl_sym = sl < 18u ? kCodeLenOrder[sl] : 31u;
        if (is_complex && sl < ncl) cl_len = br.read(5);
        if (cl_len > 9u) cl_len = 0u;                              // > 9 is invalid (2^9 table in the reference)
        // canonical code of my code-length symbol: the symbols that precede it in (length, symbol) order each take 2^(my length - theirs) of
        // its code space.  Round 5: counted from nine ballots (which lanes hold a code of length l?) -- the shorter ones by popcount, the
        // ones of my own length by popcount under "lanes whose symbol is smaller than mine" (the symbol order is fixed: kCodeLenOrder) --
        // instead of eighteen broadcasts of every lane's (length, symbol) to every lane.
        uint32_t cl_code = 0;
        if (!(kAblate & kAblTabCl)) {
            const uint32_t smaller = cl_smaller_lanes(sl);
#pragma unroll
            for (uint32_t l = 1; l <= 9u; ++l) {
                const uint32_t m = wave::half_of(wave::ballot_eq(cl_len, l));
                if (l < cl_len) cl_code += (uint32_t)__popc(m) << (cl_len - l);
                else if (l == cl_len) cl_code += (uint32_t)__popc(m & smaller);
            }
        }
        // LUT over the next `tb` stream bits (LSB-first), tb = the longest code-length code of the page (<= 9; typically 4 .. 6): entry =
        // sym << 4 | len.  (Rounds 1-4 always built the reference's 2^9 table: a lane with a 1- or 2-bit code wrote 256 or 128 entries.)
        const uint32_t cl_longest = wave::half_max(cl_len);
        const uint32_t tb = cl_longest > 0u ? cl_longest : 1u;
        if (is_complex) for (uint32_t e = sl; e < (1u << tb); e += 32u) scratch16[e] = 0;
        wave::sync();
        {
            const uint32_t reps = (is_complex && sl < 18u && cl_len) ? (1u << (tb - cl_len)) : 0u;
            const uint32_t rcode = cl_len ? (__brev(cl_code) >> (32u - cl_len)) : 0u;
            for (uint32_t m = 0; m < reps; ++m) scratch16[rcode + (m << cl_len)] = (uint16_t)((cl_sym << 4) | cl_len);
        }
        // the code lengths start out as zeros: the RLE pass below only stores the non-zero ones (most of an alphabet is unused)
        if (is_complex) for (uint32_t o = 16u * sl; o < ((A + 15u) & ~15u); o += 512u) store16(codelens + o, Bytes16{0u, 0u, 0u, 0u});
        wave::sync();

        // RLE symbols: one (plus its extra bits) per sub-stream, round-robin, until A lengths exist
        uint32_t produced = (is_complex && !(kAblate & kAblTabRle)) ? 0u : A;
        uint32_t prev_len = 8;                                     // BROTLI_INITIAL_REPEATED_CODE_LENGTH
        while (wave::any(produced < A)) {
            const bool act = produced < A;
            uint32_t sym = 0, clen = 0, run = 0, extra = 0, nextra = 0;
            if (act) {
                br.ensure(16);                                     // <= 9-bit code + up to 3 extra bits
                const uint32_t e = scratch16[br.peek(tb)];
                sym = e >> 4; clen = e & 15u;
                nextra = sym == 16u ? 2u : (sym == 17u ? 3u : 0u);
                extra = ((uint32_t)(br.buf >> clen)) & ((1u << nextra) - 1u);
                run = sym >= 16u ? 3u + extra : 1u;
            }
            const uint32_t incl = wave::half_scan_incl(run);
            const uint32_t start = produced + incl - run;
            const bool valid = act && start < A;
            if (valid) br.consume(clen + nextra);
            const uint32_t lit_mask = wave::half_ballot(valid && sym < 16u);
            const uint32_t before = lit_mask & ((1u << sl) - 1u);
            const uint32_t from_lane = wave::half_shfl(sym, before ? msb_u32(before) : 0u);
            const uint32_t last_lit = wave::half_bcast(sym, lit_mask ? msb_u32(lit_mask) : 0u);
            uint32_t value = sym;                                  // literal length
            if (sym == 17u) value = 0u;
            else if (sym == 16u) value = before ? from_lane : prev_len;    // repeat previous *literal* length
            if (valid && value != 0u) {                             // (zeros are there already)
                const uint32_t end = min_u32(start + run, A);
                for (uint32_t s = start; s < end; ++s) codelens[s] = (uint8_t)value;
            }
            // (the valid lanes are a prefix of the half, and when a lane is not valid the lengths are complete: the sum over the valid lanes
            // and the sum over all lanes give the same `produced` after the clamp -- one broadcast instead of a second scan)
            produced = min_u32(A, produced + wave::half_bcast(incl, 31u));
            if (lit_mask) prev_len = last_lit;
        }
        wave::sync();

        // canonical build.  Each lane owns a contiguous block of symbols; per-(length, lane)
        // counters give every symbol its rank without atomics.
        if (!(kAblate & kAblTabCanon)) {
        uint16_t* cnt = scratch16;                                 // [16][32]
        const uint32_t blk = (A + 31u) / 32u;
        const uint32_t b0 = sl * blk, b1 = min_u32(A, b0 + blk);
#if BROTLIG_TUNE_TABLE_UNROLL
#pragma unroll
#endif
        for (uint32_t l = 0; l < 16u; ++l) if (is_complex) cnt[l * 32u + sl] = 0;
        if (is_complex && A != kLitAlphabet) for (uint32_t w = sl; w < (sym_cap(A) + 2u) / 3u; w += 32u) t.sorted[w] = 0u;
        wave::sync();
        if (is_complex)
            for (uint32_t s = b0; s < b1; ++s) { const uint32_t l = codelens[s] & 15u; if (l) cnt[l * 32u + sl]++; }
        wave::sync();
        uint32_t code = 0, off = 0, prev_count = 0;
#if BROTLIG_TUNE_TABLE_UNROLL
#pragma unroll
#endif
        for (uint32_t l = 1; l < 16u; ++l) {
            const uint32_t c = is_complex ? cnt[l * 32u + sl] : 0u;
            const uint32_t incl = wave::half_scan_incl(c);
            const uint32_t total = wave::half_bcast(incl, 31);
            if (is_complex) cnt[l * 32u + sl] = (uint16_t)(off + incl - c);
            code = (code + prev_count) << 1;
            if (is_complex && sl == 0u) {
                t.limit[l] = (uint16_t)min_u32((code + total) << (15u - l), 32768u);
                t.first_offs[l] = min_u32(code << (15u - l), 32768u) | (off << 16);
            }
            off += total; prev_count = total;
        }
        wave::sync();
        if (is_complex)
            for (uint32_t s = b0; s < b1; ++s) {
                const uint32_t l = codelens[s] & 15u;
                if (l) { const uint32_t p = cnt[l * 32u + sl]++; table_set_sym(t, min_u32(p, A - 1u), s); }
            }
        // symbols beyond the LDS arrays went to global memory: stores first, then the reads below and in the rounds
        // (same CU, same L1: workgroup scope is enough)
        }
        if (A != kLitAlphabet) wave::global_fence(); else wave::sync();
        // primary LUT.  Round 5: filled in CODE order -- lane sl owns the lut_size / 32 consecutive code prefixes from (lut_size / 32) * sl on,
        // entry index = the prefix bit-reversed.  The length of a code is monotone in its left-justified value (the limits are: each is the
        // previous one plus the codes of its length, clamped), so only a lane's first prefix takes the search over all fifteen limits; from one
        // prefix to the next the length is walked up against limit[l].  (Rounds 1-4: index order, two fifteen-compare searches per entry.)
        if (is_complex && !(kAblate & kAblTabLut)) {
            // length of the code whose left-justified 15-bit value range contains v (16: none): the limits are monotone, so a binary search
            // over limit[1..15] (four reads) finds the first one above v
            auto length_of = [&t](uint32_t v) {
                uint32_t l = 0;                                     // invariant: limit[l] <= v (limit[0] taken as 0), answer in (l, l + span]
                l += v >= (uint32_t)t.limit[l + 8u] ? 8u : 0u;
                l += v >= (uint32_t)t.limit[l + 4u] ? 4u : 0u;
                l += v >= (uint32_t)t.limit[l + 2u] ? 2u : 0u;
                l += v >= (uint32_t)t.limit[l + 1u] ? 1u : 0u;
                return l + 1u;
            };
            const uint32_t lut_bits = (uint32_t)t.lut_bits, per = lut_size >> 5, step = 1u << (15u - lut_bits);
            uint32_t v = (per * sl) << (15u - lut_bits);           // left-justified 15-bit value of my first prefix
            uint32_t l = length_of(v);
            uint32_t lim_l = l <= 15u ? (uint32_t)t.limit[l] : 0xFFFFFFFFu;
            uint32_t fo = l <= 15u ? t.first_offs[l] : 0u;
            for (uint32_t i = 0; i < per; ++i, v += step) {
                if (v >= lim_l) {
                    do { ++l; lim_l = l <= 15u ? (uint32_t)t.limit[l] : 0xFFFFFFFFu; } while (v >= lim_l);
                    fo = l <= 15u ? t.first_offs[l] : 0u;
                }
                uint32_t entry = kLongCode;
                if (l <= 15u) {
                    const uint32_t idx = (fo >> 16) + ((v - (fo & 0xFFFFu)) >> (15u - l));
                    if (l <= lut_bits) {
                        entry = (table_sym(t, min_u32(idx, A - 1u)) << 4) | l;
                    } else if (v + step - 1u < lim_l && idx + (1u << (l - lut_bits)) <= A) {
                        // every code under this prefix has length l: they are consecutive in code order, so the
                        // symbol is sorted[idx + the next l - lut_bits code bits] -- no length search at decode time
                        entry = kLutSubtree | (idx << 4) | l;
                    }
                }
                t.lut[__brev(per * sl + i) >> (32u - lut_bits)] = (uint16_t)entry;
            }
        }
    }
    // -- trivial / simple LUTs (written last: the complex path uses LUT areas as scratch)
    if (is_trivial) {
        for (uint32_t e = sl; e < lut_size; e += 32u) t.lut[e] = (uint16_t)(s0 << 4);
    } else if (is_simple) {
        const uint32_t shape = nsym < 4u ? nsym - 2u : (tree_select ? 3u : 2u);   // BrotligHuffmanTable.cpp:26-38
        for (uint32_t e = sl; e < lut_size; e += 32u) {
            const uint32_t b0 = e & 1u, b1 = (e >> 1) & 1u, b2 = (e >> 2) & 1u;
            uint32_t k, len;
            if (shape == 0u) { k = b0; len = 1u; }
            else if (shape == 1u) { k = b0 ? 1u + b1 : 0u; len = b0 ? 2u : 1u; }
            else if (shape == 2u) { k = b0 * 2u + b1; len = 2u; }
            else { k = !b0 ? 0u : (!b1 ? 1u : 2u + b2); len = !b0 ? 1u : (!b1 ? 2u : 3u); }
            const uint32_t sym = k == 0u ? s0 : k == 1u ? s1 : k == 2u ? s2 : s3;
            t.lut[e] = (uint16_t)((sym << 4) | len);
        }
    }
    wave::sync();
    return defined;
}

}  // namespace brotlig
