// brotlig_encoder.cpp -- functional Brotli-G v1.1 stream encoder (host only).
//
// Role: input generator for the decoder tests and for bench.py.  The reference encoder
// (src/BrotligEncoder.cpp, src/encoder/PageEncoder.cpp) is built on google/brotli v1.0.9
// internals (Zopfli parse, histogram clustering) that are not available here, and decode
// parity does not depend on which valid stream is decoded, so this is an independent
// encoder that emits the same container and page format:
//   container      : inc/DataStream.h:28-108, src/BrotligEncoder.cpp:575-611
//   page layout    : src/common/BrotligSwizzler.cpp:68-189 (header, size table, 32 sub-streams)
//   prefix codes   : src/encoder/BrotligHuffman.cpp:262-364 (trivial / simple / complex + RLE)
//   round layout   : src/encoder/PageEncoder.cpp:475-540 (commands, then redistributed literals)
//   conditioning   : src/common/BrotligDataConditioner.cpp:28-119, PageEncoder.cpp:576-612
// It is NOT part of the decode product path.
#include <algorithm>
#include <atomic>
#include <thread>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

#include "brotlig_encoder.h"

namespace {

constexpr uint32_t kNumStreams = 32;
constexpr uint32_t kIcpAlphabet = 728, kDistAlphabet = 544, kLitAlphabet = 256;

const uint32_t kInsBase[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98,
                               130, 194, 322, 578, 1090, 2114, 6210, 22594};
const uint32_t kInsExtra[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
const uint32_t kCopyBase[24] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54,
                                70, 102, 134, 198, 326, 582, 1094, 2118};
const uint32_t kCopyExtra[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};

inline uint32_t bit_width(uint32_t x) { return x ? 32u - (uint32_t)__builtin_clz(x) : 0u; }

uint32_t code_of(const uint32_t* base, uint32_t v)
{
    uint32_t c = 0;
    while (c + 1 < 24 && base[c + 1] <= v) ++c;
    return c;
}

// ---------------------------------------------------------------------------------------
// LSB-first bit writer (one per sub-stream)
struct BitWriter {
    std::vector<uint8_t> bytes;
    uint64_t acc = 0;
    uint32_t nacc = 0;
    uint64_t total_bits = 0;
    void put(uint32_t v, uint32_t n)
    {
        if (!n) return;
        acc |= (uint64_t)(n == 32 ? v : (v & ((1u << n) - 1u))) << nacc;
        nacc += n; total_bits += n;
        while (nacc >= 8) { bytes.push_back((uint8_t)acc); acc >>= 8; nacc -= 8; }
    }
    void flush() { if (nacc) { bytes.push_back((uint8_t)acc); acc = 0; nacc = 0; } }
};

inline uint32_t reverse_bits(uint32_t v, uint32_t n)
{
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

// ---------------------------------------------------------------------------------------
// Length-limited Huffman code lengths (count-floor doubling until the tree fits, the scheme
// google/brotli's BrotliCreateHuffmanTree uses).
void huffman_lengths(const std::vector<uint32_t>& counts, uint32_t limit, std::vector<uint8_t>& depth)
{
    const size_t n = counts.size();
    depth.assign(n, 0);
    std::vector<uint32_t> used;
    for (size_t i = 0; i < n; ++i) if (counts[i]) used.push_back((uint32_t)i);
    if (used.empty()) return;
    if (used.size() == 1) { depth[used[0]] = 1; return; }
    struct Node { uint64_t w; int l, r; };
    for (uint32_t floor_ = 1;; floor_ *= 2) {
        std::vector<Node> nodes;
        nodes.reserve(2 * used.size());
        for (uint32_t s : used) nodes.push_back({std::max<uint64_t>(counts[s], floor_), -1, (int)s});
        // sort leaves ascending by weight (stable on symbol for determinism)
        std::vector<int> leaves(used.size());
        for (size_t i = 0; i < leaves.size(); ++i) leaves[i] = (int)i;
        std::stable_sort(leaves.begin(), leaves.end(), [&](int a, int b) { return nodes[a].w < nodes[b].w; });
        std::vector<int> q2;            // internal nodes, created in non-decreasing weight order
        size_t i1 = 0, i2 = 0;
        auto pop = [&]() -> int {
            if (i1 < leaves.size() && (i2 >= q2.size() || nodes[leaves[i1]].w <= nodes[q2[i2]].w)) return leaves[i1++];
            return q2[i2++];
        };
        const size_t nleaves = used.size();
        for (size_t k = 0; k + 1 < nleaves; ++k) {
            int a = pop(), b = pop();
            nodes.push_back({nodes[a].w + nodes[b].w, a, b});
            q2.push_back((int)nodes.size() - 1);
        }
        // depth-first assignment
        std::vector<std::pair<int, uint32_t>> st;
        st.push_back({(int)nodes.size() - 1, 0});
        uint32_t maxd = 0;
        while (!st.empty()) {
            auto [id, d] = st.back(); st.pop_back();
            if (nodes[id].l < 0) { depth[nodes[id].r] = (uint8_t)std::min<uint32_t>(d, 255); maxd = std::max(maxd, d); }
            else { st.push_back({nodes[id].l, d + 1}); st.push_back({nodes[id].r, d + 1}); }
        }
        if (maxd <= limit) return;
    }
}

// canonical codes: ascending length, then ascending symbol (src/decoder/BrotligHuffmanTable.cpp:44-71)
void canonical_codes(const std::vector<uint8_t>& depth, std::vector<uint16_t>& code)
{
    uint32_t cnt[17] = {0}, next[17] = {0};
    for (uint8_t d : depth) if (d) ++cnt[d];
    for (uint32_t l = 1; l <= 16; ++l) next[l] = (next[l - 1] + cnt[l - 1]) << 1;
    code.assign(depth.size(), 0);
    for (size_t s = 0; s < depth.size(); ++s) if (depth[s]) code[s] = (uint16_t)next[depth[s]]++;
}

struct PrefixCode {
    std::vector<uint8_t> depth;     // 0 = unused; for trivial codes every depth is 0
    std::vector<uint16_t> code;     // MSB-first canonical value (written bit-reversed)
    void put(BitWriter& w, uint32_t sym) const { if (depth[sym]) w.put(reverse_bits(code[sym], depth[sym]), depth[sym]); }
};

struct Streams {
    BitWriter w[kNumStreams];
    uint32_t cur = 0;
    BitWriter& at() { return w[cur]; }
    void next() { cur = (cur + 1) & (kNumStreams - 1); }
    void reset() { cur = 0; }
};

// Code-length RLE tokens (mirror of src/common/BrotligUtils.cpp:76-228 as read back by
// src/decoder/BrotligHuffmanTable.cpp:163-195).
struct Token { uint8_t sym; uint8_t extra; };
// `corners`: the token sequences the reference's DECODER accepts and its encoder never chooses (BrotligHuffmanTable.cpp:163-195: every
// literal token -- a 0 too -- becomes the length that 16 repeats, 16 and 17 leave it alone): a non-zero run that follows a zero run coded
// with 17 goes straight to 16 when the last literal still has its length, and every other zero run is a literal 0 followed by 16s.
void rle_tokens(const std::vector<uint8_t>& depth, bool use_rle, std::vector<Token>& out, bool corners = false)
{
    out.clear();
    const size_t n = depth.size();
    int prev_literal = 8;       // decoder's initial "previous" (BrotligHuffmanTable.cpp:149)
    bool last_was_zero_run = false;
    bool first = true;
    uint32_t zero_runs = 0;
    size_t i = 0;
    while (i < n) {
        uint8_t v = depth[i];
        size_t r = 1;
        while (i + r < n && depth[i + r] == v) ++r;
        size_t left = r;
        if (!use_rle) {
            for (; left; --left) out.push_back({v, 0});
            prev_literal = v; last_was_zero_run = false; first = false;
        } else if (v == 0) {
            if (first) { out.push_back({0, 0}); --left; prev_literal = 0; first = false; last_was_zero_run = false; }
            if (corners && (zero_runs++ & 1u) && left >= 4) {           // 0, then 16s repeating it
                if (prev_literal != 0) { out.push_back({0, 0}); --left; prev_literal = 0; }
                while (left >= 3) { size_t k = std::min<size_t>(left, 6); out.push_back({16, (uint8_t)(k - 3)}); left -= k; }
                last_was_zero_run = false;
            }
            while (left >= 3) { size_t k = std::min<size_t>(left, 10); out.push_back({17, (uint8_t)(k - 3)}); left -= k; last_was_zero_run = true; }
            for (; left; --left) { out.push_back({0, 0}); prev_literal = 0; last_was_zero_run = false; }
        } else {
            if (first || prev_literal != v || (last_was_zero_run && !corners)) { out.push_back({v, 0}); --left; prev_literal = v; first = false; }
            last_was_zero_run = false;
            while (left >= 3) { size_t k = std::min<size_t>(left, 6); out.push_back({16, (uint8_t)(k - 3)}); left -= k; }
            for (; left; --left) out.push_back({v, 0});
        }
        i += r;
    }
}

// Smooth population counts so that the code lengths built from them form longer runs (fewer tokens in the
// description).  In the spirit of brotli's BrotliOptimizeHuffmanCountsForRle (which the reference calls,
// src/encoder/PageEncoder.cpp:430-437), restated freely: stretches of existing long runs are left alone;
// elsewhere, neighbouring counts that stay within a band around their running mean are replaced by that mean.
// A symbol that occurs keeps a non-zero count; symbols inside a smoothed stretch may gain one.
void smooth_counts_for_rle(std::vector<uint32_t>& c)
{
    size_t n = c.size();
    while (n > 0 && c[n - 1] == 0) --n;                                // trailing zeros stay zeros
    size_t nonzero = 0;
    for (size_t i = 0; i < n; ++i) nonzero += c[i] != 0;
    if (nonzero < 16) return;
    // runs that already encode well: >= 5 zeros, or >= 7 equal non-zero counts
    std::vector<uint8_t> keep(n, 0);
    for (size_t i = 0; i < n;) {
        size_t j = i;
        while (j < n && c[j] == c[i]) ++j;
        if ((c[i] == 0 && j - i >= 5) || (c[i] != 0 && j - i >= 7)) for (size_t k = i; k < j; ++k) keep[k] = 1;
        i = j;
    }
    auto close = [&](size_t from, size_t to, uint64_t sum) {          // [from, to): replace by the mean
        const size_t len = to - from;
        if (len < 4 && !(len == 3 && sum == 0)) return;
        uint32_t mean = (uint32_t)((sum + len / 2) / len);
        if (mean == 0 && sum != 0) mean = 1;
        for (size_t k = from; k < to; ++k) c[k] = (sum == 0) ? 0u : std::max<uint32_t>(mean, 1u);
    };
    size_t from = 0; uint64_t sum = 0;
    for (size_t i = 0; i <= n; ++i) {
        bool brk = i == n || keep[i] || (i > from && keep[i - 1]);
        if (!brk && i > from) {
            // band: within ~40 % + 2 of the stretch's mean so far; zeros and non-zeros do not mix
            const double mean = (double)sum / (double)(i - from);
            const double lo = mean * 0.6 - 2.0, hi = mean * 1.4 + 2.0;
            if ((double)c[i] < lo || (double)c[i] > hi || ((c[i] == 0) != (sum == 0))) brk = true;
        }
        if (brk) {
            if (i > from) close(from, i, sum);
            from = i; sum = 0;
            if (i < n && keep[i]) { from = i + 1; continue; }
        }
        if (i < n) sum += c[i];
    }
}

// Emit one prefix-code description (ICP / distance / literal) and return the code to use.
PrefixCode emit_prefix_code(Streams& S, const std::vector<uint32_t>& hist, uint32_t flags)
{
    const uint32_t alphabet = (uint32_t)hist.size();
    const uint32_t maxbits = bit_width(alphabet - 1);
    PrefixCode pc;
    pc.depth.assign(alphabet, 0); pc.code.assign(alphabet, 0);
    std::vector<uint32_t> used;
    for (uint32_t s = 0; s < alphabet; ++s) if (hist[s]) used.push_back(s);
    S.reset();
    const bool force_complex = (flags & BROTLIG_ENC_FORCE_COMPLEX_TABLES) != 0;
    const bool corners = (flags & BROTLIG_ENC_DECODER_CORNERS) != 0;
    if (used.size() <= 1) {                                            // trivial
        S.at().put(0, 2); S.at().put(corners ? 15 : 0, 4);              // (the four bits the reader skips, BrotligHuffmanTable.cpp:87)
        S.at().put(used.empty() ? 0 : used[0], maxbits);
        S.reset();
        return pc;
    }
    if (used.size() <= 4 && !force_complex) {                          // simple
        std::vector<uint32_t> ord = used;
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return hist[a] > hist[b]; });
        static const uint8_t lens[4][4] = {{1, 1, 0, 0}, {1, 2, 2, 0}, {2, 2, 2, 2}, {1, 2, 3, 3}};
        static const uint16_t codes[4][4] = {{0, 1, 0, 0}, {0, 2, 3, 0}, {0, 1, 2, 3}, {0, 2, 6, 7}};
        uint32_t nsym = (uint32_t)ord.size(), sel = 0, idx;
        if (nsym == 4) {
            uint64_t c0 = 2ull * (hist[ord[0]] + hist[ord[1]] + hist[ord[2]] + hist[ord[3]]);
            uint64_t c1 = 1ull * hist[ord[0]] + 2ull * hist[ord[1]] + 3ull * (hist[ord[2]] + hist[ord[3]]);
            sel = c1 < c0;
        }
        idx = nsym < 4 ? nsym - 2 : (sel ? 3 : 2);
        // list order = (length, symbol); lengths follow frequency rank
        std::vector<std::pair<uint8_t, uint32_t>> lst;
        for (uint32_t k = 0; k < nsym; ++k) lst.push_back({lens[idx][k], ord[k]});
        std::sort(lst.begin(), lst.end());
        // (the reader gives the k-th listed symbol the k-th code of the shape, BrotligHuffmanTable.cpp:100-116: symbols of one length may come in
        // any order -- descending with `corners` --, and the sixth header bit is skipped)
        if (corners) std::sort(lst.begin(), lst.end(), [](const std::pair<uint8_t, uint32_t>& a, const std::pair<uint8_t, uint32_t>& b) {
            return a.first != b.first ? a.first < b.first : a.second > b.second; });
        S.at().put(1, 2); S.at().put(nsym - 1, 2); S.at().put(sel, 1); S.at().put(corners ? 1 : 0, 1);
        for (uint32_t k = 0; k < nsym; ++k) {
            S.at().put(lst[k].second, maxbits);
            pc.depth[lst[k].second] = lens[idx][k];
            pc.code[lst[k].second] = codes[idx][k];
            S.next();
        }
        S.reset();
        return pc;
    }
    // complex.  plan(): code lengths from `counts`, their RLE tokens and the code-length code; returns the bits the
    // description plus the coded symbols (true counts) will take.
    std::vector<Token> toks;
    std::vector<uint8_t> tdepth; std::vector<uint16_t> tcode;
    auto plan = [&](const std::vector<uint32_t>& counts, std::vector<uint8_t>& depth, std::vector<Token>& tk,
                    std::vector<uint8_t>& td, std::vector<uint16_t>& tc) -> uint64_t {
        huffman_lengths(counts, 15, depth);
        if (used.size() == 1) { depth[used[0]] = 1; }                  // only reachable with force_complex
        if (used.size() < 2) {                                         // a 1-symbol complex code is incomplete: add a dummy
            uint32_t other = used[0] == 0 ? 1 : 0;
            depth[other] = 1;
        }
        rle_tokens(depth, !(flags & BROTLIG_ENC_NO_CODELEN_RLE), tk, (flags & BROTLIG_ENC_DECODER_CORNERS) != 0);
        std::vector<uint32_t> thist(18, 0);
        for (auto& t : tk) ++thist[t.sym];
        uint32_t distinct = 0; for (uint32_t c : thist) distinct += c != 0;
        if (distinct < 2) {                                            // Appendix D.6: never a 1-symbol code-length code
            rle_tokens(depth, true, tk, (flags & BROTLIG_ENC_DECODER_CORNERS) != 0);
            std::fill(thist.begin(), thist.end(), 0);
            for (auto& t : tk) ++thist[t.sym];
        }
        huffman_lengths(thist, 7, td);                                 // <= 7: BrotliGCompute.hlsl:58
        canonical_codes(td, tc);
        uint64_t bits = 6 + 18 * 5;
        for (auto& t : tk) bits += td[t.sym] + (t.sym == 16 ? 2 : t.sym == 17 ? 3 : 0);
        for (uint32_t sy = 0; sy < alphabet; ++sy) bits += (uint64_t)hist[sy] * depth[sy];
        return bits;
    };
    uint64_t best_bits = plan(hist, pc.depth, toks, tdepth, tcode);
    if (flags & BROTLIG_ENC_SMOOTH_HISTOGRAMS) {
        // the reference smooths the counts before building the code (src/encoder/PageEncoder.cpp:430-437 calls
        // brotli's BrotliOptimizeHuffmanCountsForRle); here the smoothed counts are kept only when they pay
        std::vector<uint32_t> smooth = hist;
        smooth_counts_for_rle(smooth);
        std::vector<uint8_t> d2, td2; std::vector<Token> tk2; std::vector<uint16_t> tc2;
        const uint64_t b2 = plan(smooth, d2, tk2, td2, tc2);
        if (b2 < best_bits) { best_bits = b2; pc.depth = d2; toks = tk2; tdepth = td2; tcode = tc2; }
    }
    canonical_codes(pc.depth, pc.code);
    static const uint8_t order[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
    S.at().put(2, 2); S.at().put(18 - 4, 4);
    for (uint32_t k = 0; k < 18; ++k) { S.at().put(tdepth[order[k]], 5); S.next(); }
    S.reset();
    for (auto& t : toks) {
        S.at().put(reverse_bits(tcode[t.sym], tdepth[t.sym]), tdepth[t.sym]);
        if (t.sym == 16) S.at().put(t.extra, 2);
        else if (t.sym == 17) S.at().put(t.extra, 3);
        S.next();
    }
    S.reset();
    return pc;
}

// ---------------------------------------------------------------------------------------
// LZ77 parse
struct Command {
    uint32_t insert_len, copy_len, dist;
    // derived
    uint16_t icp_sym; uint16_t dist_sym; uint32_t extra_bits_val; uint32_t extra_nbits; // insert|copy extras
    uint32_t dist_extra, dist_nbits; bool has_dist_sym;
};

struct Parser {
    const uint8_t* d; uint32_t n; uint32_t max_chain; bool lazy; bool use_ring;
    std::vector<int32_t> head, prev;
    static constexpr uint32_t kHashBits = 15;
    uint32_t hash4(uint32_t p) const { uint32_t v; memcpy(&v, d + p, 4); return (v * 0x9E3779B1u) >> (32 - kHashBits); }
    uint32_t match_len(uint32_t a, uint32_t b, uint32_t maxlen) const
    {
        uint32_t l = 0;
        while (l + 8u <= maxlen) {                                  // eight bytes a step (little-endian: the first mismatch is the lowest set bit)
            uint64_t x, y; memcpy(&x, d + a + l, 8); memcpy(&y, d + b + l, 8);
            if (x != y) return l + ((uint32_t)__builtin_ctzll(x ^ y) >> 3);
            l += 8u;
        }
        while (l < maxlen && d[a + l] == d[b + l]) ++l;
        return l;
    }
    void insert(uint32_t p) { if (p + 4 <= n) { uint32_t h = hash4(p); prev[p] = head[h]; head[h] = (int32_t)p; } }
    // best match at p given the current ring; returns length (0 = none)
    uint32_t find(uint32_t p, const uint32_t ring[4], uint32_t& best_dist) const
    {
        uint32_t best = 0; int best_score = 0; best_dist = 0;
        const uint32_t maxlen = n - p;
        auto consider = [&](uint32_t dist, uint32_t len, int dist_cost) {
            int score = (int)len * 8 - dist_cost;
            if (len >= 2 && score > best_score) { best_score = score; best = len; best_dist = dist; }
        };
        if (use_ring) {
            for (int r = 0; r < 4; ++r) {
                uint32_t dist = ring[r];
                if (dist == 0 || dist > p) continue;
                uint32_t l = match_len(p - dist, p, maxlen);
                if (l >= (r == 0 ? 2u : 3u)) consider(dist, l, r == 0 ? 4 : 10);
            }
        }
        if (p + 4 <= n) {
            int32_t c = head[hash4(p)];
            for (uint32_t it = 0; c >= 0 && it < max_chain; ++it, c = prev[c]) {
                uint32_t dist = p - (uint32_t)c;
                uint32_t l = match_len((uint32_t)c, p, maxlen);
                if (l >= 4) consider(dist, l, 14 + (int)bit_width(dist));
                if (l == maxlen) break;
            }
        }
        return best;
    }
};

void parse_page(const uint8_t* data, uint32_t n, const BrotligEncodeOptions& o, std::vector<Command>& cmds,
                std::vector<uint8_t>& literals)
{
    cmds.clear(); literals.clear();
    Parser P; P.d = data; P.n = n;
    P.max_chain = o.max_chain ? o.max_chain : 24;
    P.lazy = !(o.flags & BROTLIG_ENC_NO_LAZY);
    P.use_ring = !(o.flags & BROTLIG_ENC_NO_RING_CODES);
    P.head.assign(1u << Parser::kHashBits, -1); P.prev.assign(n, -1);
    uint32_t ring[4] = {4, 11, 15, 16};
    uint32_t pos = 0, lit_start = 0;
    const bool no_matches = (o.flags & BROTLIG_ENC_LITERALS_ONLY) != 0;
    while (pos < n) {
        uint32_t dist = 0, len = no_matches ? 0 : P.find(pos, ring, dist);
        if (len >= 2 && P.lazy && pos + 1 < n && len < 64) {
            P.insert(pos);
            uint32_t d2 = 0, l2 = P.find(pos + 1, ring, d2);
            if (l2 > len + 1) { ++pos; continue; }                     // take a literal, retry at pos+1
            // keep current match; pos already inserted
            for (uint32_t k = 1; k < len; ++k) P.insert(pos + k);
        } else if (len >= 2) {
            for (uint32_t k = 0; k < len; ++k) P.insert(pos + k);
        } else {
            P.insert(pos); ++pos; continue;
        }
        Command c{}; c.insert_len = pos - lit_start; c.copy_len = len; c.dist = dist;
        literals.insert(literals.end(), data + lit_start, data + pos);
        cmds.push_back(c);
        // ring bookkeeping happens in assign_symbols; the parser only needs an approximation,
        // but keep it exact so ring candidates line up with what will be coded.
        if (dist != ring[0]) { ring[3] = ring[2]; ring[2] = ring[1]; ring[1] = ring[0]; ring[0] = dist; }
        pos += len; lit_start = pos;
    }
    if (lit_start < n) {
        Command c{}; c.insert_len = n - lit_start; c.copy_len = 0; c.dist = 0;
        literals.insert(literals.end(), data + lit_start, data + n);
        cmds.push_back(c);
    }
}

// Distance code of `dist` given the ring and the page's NPOSTFIX / NDIRECT (PageDecoder.cpp:345-404).
inline int distance_code(uint32_t dist, const uint32_t ring[4], uint32_t npostfix, uint32_t ndirect, bool use_ring,
                         uint32_t& nbits, uint32_t& extra)
{
    nbits = 0; extra = 0;
    if (use_ring) {
        if (dist == ring[0]) return 0;
        if (dist == ring[1]) return 1;
        if (dist == ring[2]) return 2;
        if (dist == ring[3]) return 3;
        static const int add[6] = {-1, 1, -2, 2, -3, 3};
        for (int k = 0; k < 6; ++k) if ((int64_t)ring[0] + add[k] > 0 && (uint32_t)((int64_t)ring[0] + add[k]) == dist) return 4 + k;
        for (int k = 0; k < 6; ++k) if ((int64_t)ring[1] + add[k] > 0 && (uint32_t)((int64_t)ring[1] + add[k]) == dist) return 10 + k;
    }
    if (dist <= ndirect) return (int)(15 + dist);
    const uint32_t x = dist - ndirect - 1;
    const uint32_t l = x & ((1u << npostfix) - 1);
    const uint32_t y = (x >> npostfix) + 4;
    nbits = bit_width(y) - 2;
    const uint32_t p = (y >> nbits) & 1;
    const uint32_t h = 2 * (nbits - 1) + p;
    extra = y - ((2 + p) << nbits);
    return (int)(16 + ndirect + ((h << npostfix) | l));
}
// Insert-and-copy symbol of (insert code, copy code); `implicit` = distance code 0 folded into the symbol.
inline uint16_t icp_symbol(uint32_t ic, uint32_t cc, bool implicit)
{
    const uint32_t b = (cc & 7) | ((ic & 7) << 3);
    if (implicit) return (uint16_t)(b | (cc >= 8 ? 64 : 0));
    const uint32_t off = 2 * ((cc >> 3) + 3 * (ic >> 3));
    return (uint16_t)(((off << 5) + 0x40 + ((0x520D40u >> off) & 0xC0)) | b);
}

// distance -> (symbol, extra) with the decoder's ring semantics (PageDecoder.cpp:345-404)
void assign_symbols(std::vector<Command>& cmds, uint32_t npostfix, uint32_t ndirect, bool use_ring)
{
    uint32_t ring[4] = {4, 11, 15, 16};
    for (size_t ci = 0; ci < cmds.size(); ++ci) {
        Command& c = cmds[ci];
        uint32_t ic = code_of(kInsBase, c.insert_len);
        c.dist_nbits = 0; c.dist_extra = 0; c.has_dist_sym = false; c.dist_sym = 0;
        if (c.copy_len == 0) {                                         // insert-only tail (PageEncoder.cpp:130-142)
            c.icp_sym = (uint16_t)(704 + ic);
            c.extra_bits_val = c.insert_len - kInsBase[ic]; c.extra_nbits = kInsExtra[ic];
            continue;
        }
        uint32_t cc = code_of(kCopyBase, c.copy_len);
        const int code = distance_code(c.dist, ring, npostfix, ndirect, use_ring, c.dist_nbits, c.dist_extra);
        const bool implicit = code == 0 && ic < 8 && cc < 16;
        c.icp_sym = icp_symbol(ic, cc, implicit);
        if (!implicit) { c.has_dist_sym = true; c.dist_sym = (uint16_t)code; }
        c.extra_nbits = kInsExtra[ic] + kCopyExtra[cc];
        // insert extra in the low bits, copy extra above; both fields <= 24 bits so emit separately
        c.extra_bits_val = 0;   // emitted field-wise in emit_command
        if (code != 0) { ring[3] = ring[2]; ring[2] = ring[1]; ring[1] = ring[0]; ring[0] = c.dist; }
    }
}

// ---------------------------------------------------------------------------------------
// Optimal parse (opt-in, BROTLIG_ENC_OPTIMAL_PARSE): shortest path over the page's positions under a
// bit-cost model taken from a previous parse of the same page -- the idea of the reference's
// Zopfli-style pass (src/encoder/PageEncoder.cpp:87-147 drives brotli's backward-references code),
// restated here in a simple form: a node is a command boundary, an edge is one command
// (insert run + copy), priced as -log2 of the symbol frequencies plus extra bits.
struct CostModel {
    float lit[256], icp[kIcpAlphabet], dist[kDistAlphabet];
    void from_histograms(const std::vector<uint32_t>& hl, const std::vector<uint32_t>& hi, const std::vector<uint32_t>& hd)
    {
        auto fill = [](const std::vector<uint32_t>& h, float* out, size_t n) {
            double total = 0; for (size_t k = 0; k < n; ++k) total += h[k];
            const double missing = std::log2(total + 2.0) + 1.0;        // a symbol the previous parse never used
            // a prefix code spends at least one bit per symbol, however frequent
            for (size_t k = 0; k < n; ++k) out[k] = h[k] ? std::max(1.0f, (float)std::log2(total / h[k])) : (float)missing;
        };
        fill(hl, lit, 256); fill(hi, icp, kIcpAlphabet); fill(hd, dist, kDistAlphabet);
    }
};

struct PathNode {
    float cost;                 // cheapest way to have a command boundary here
    uint32_t from, ins, len, dist;
    uint32_t ring[4];           // distance ring after the command that ends here
};

void parse_page_optimal(const uint8_t* data, uint32_t n, const BrotligEncodeOptions& o, uint32_t npostfix, uint32_t ndirect,
                        const CostModel& cm, std::vector<Command>& cmds, std::vector<uint8_t>& literals)
{
    const bool use_ring = !(o.flags & BROTLIG_ENC_NO_RING_CODES);
    const uint32_t max_chain = o.max_chain ? o.max_chain : 24;
    Parser P; P.d = data; P.n = n; P.max_chain = max_chain; P.lazy = false; P.use_ring = use_ring;
    P.head.assign(1u << Parser::kHashBits, -1); P.prev.assign(n, -1);
    std::vector<float> litcum(n + 1, 0.f);
    for (uint32_t i = 0; i < n; ++i) litcum[i + 1] = litcum[i] + cm.lit[data[i]];
    const float kInf = 1e30f;
    std::vector<PathNode> node(n + 1);
    for (auto& nd : node) { nd.cost = kInf; nd.from = 0; nd.ins = nd.len = nd.dist = 0; }
    node[0].cost = 0; node[0].ring[0] = 4; node[0].ring[1] = 11; node[0].ring[2] = 15; node[0].ring[3] = 16;
    // the few cheapest boundaries to start the next command from: key = cost - literal cost up to there
    constexpr int kStarts = 4;
    uint32_t starts[kStarts]; int nstarts = 0;
    auto key = [&](uint32_t i) { return node[i].cost - litcum[i]; };
    auto push_start = [&](uint32_t i) {                                 // `starts` stays sorted by key, cheapest first
        int k;
        if (nstarts < kStarts) k = nstarts++;
        else if (key(i) < key(starts[kStarts - 1])) k = kStarts - 1;
        else return;
        starts[k] = i;
        while (k > 0 && key(starts[k - 1]) > key(starts[k])) { std::swap(starts[k - 1], starts[k]); --k; }
    };
    // Edge costs are split into what depends on the start (insert code, literals), on start and distance (distance
    // symbol) and on the copy length (copy code, the ICP symbol), so that walking the lengths of one match from one
    // start is a table lookup and a compare per length.
    static const std::vector<uint8_t> copy_code = [] {
        std::vector<uint8_t> t(2200);
        for (uint32_t l = 0; l < t.size(); ++l) t[l] = (uint8_t)code_of(kCopyBase, l < 2 ? 2 : l);
        return t;
    }();
    auto cc_of = [&](uint32_t len) { return len < copy_code.size() ? (uint32_t)copy_code[len] : code_of(kCopyBase, len); };
    std::vector<float> cmd_cost(24 * 24 * 2);                           // [insert code][copy code][implicit]: ICP symbol + copy extra bits
    for (uint32_t ic = 0; ic < 24; ++ic)
        for (uint32_t cc = 0; cc < 24; ++cc)
            for (uint32_t im = 0; im < 2; ++im)
                cmd_cost[(ic * 24 + cc) * 2 + im] = (im && !(ic < 8 && cc < 16)) ? kInf
                                                    : cm.icp[icp_symbol(ic, cc, im != 0)] + (float)kCopyExtra[cc];
    auto relax_range = [&](uint32_t i, uint32_t p, uint32_t lmin, uint32_t lmax, uint32_t dist) {
        const PathNode& a = node[i];
        const uint32_t ins = p - i;
        const uint32_t ic = code_of(kInsBase, ins);
        uint32_t nb, ex;
        const int code = distance_code(dist, a.ring, npostfix, ndirect, use_ring, nb, ex);
        const float base = a.cost + (litcum[p] - litcum[i]) + (float)kInsExtra[ic];
        const float dcost = cm.dist[code] + (float)nb;
        const bool can_implicit = code == 0 && ic < 8;
        const float* row = &cmd_cost[ic * 24 * 2];
        for (uint32_t len = lmin; len <= lmax; ++len) {
            const uint32_t cc = cc_of(len);
            const bool implicit = can_implicit && cc < 16;
            const float c = base + row[cc * 2 + (implicit ? 1 : 0)] + (implicit ? 0.f : dcost);
            PathNode& b = node[p + len];
            if (c < b.cost) {
                b.cost = c; b.from = i; b.ins = ins; b.len = len; b.dist = dist;
                if (code != 0) { b.ring[0] = dist; b.ring[1] = a.ring[0]; b.ring[2] = a.ring[1]; b.ring[3] = a.ring[2]; }
                else { b.ring[0] = a.ring[0]; b.ring[1] = a.ring[1]; b.ring[2] = a.ring[2]; b.ring[3] = a.ring[3]; }
            }
        }
    };
    for (uint32_t p = 0; p < n; ++p) {
        if (node[p].cost < kInf) push_start(p);
        if (nstarts == 0) { P.insert(p); continue; }
        const uint32_t maxlen = n - p;
        // candidate matches at p: ring distances of the best start, then the hash chain (nearest first,
        // only strictly longer ones are kept)
        struct M { uint32_t len, dist; } cand[48]; int nc = 0;
        if (use_ring) {
            const uint32_t* r = node[starts[0]].ring;
            for (int k = 0; k < 4; ++k) {
                if (r[k] == 0 || r[k] > p) continue;
                const uint32_t l = P.match_len(p - r[k], p, maxlen);
                if (l >= 2 && nc < 48) cand[nc++] = {l, r[k]};
            }
        }
        uint32_t best_chain = 3;
        if (p + 4 <= n) {
            int32_t c = P.head[P.hash4(p)];
            for (uint32_t it = 0; c >= 0 && it < max_chain && nc < 48; ++it, c = P.prev[c]) {
                const uint32_t l = P.match_len((uint32_t)c, p, maxlen);
                if (l > best_chain) { best_chain = l; cand[nc++] = {l, p - (uint32_t)c}; if (l == maxlen) break; }
            }
        }
        // Every copy length is priced once per start, with the first candidate that reaches it: the ring distances
        // come first (cheapest codes), then the chain's matches from the nearest on, each strictly longer than the
        // one before -- a farther match only contributes the lengths the nearer ones could not (the rule of brotli's
        // shortest-path parse, which the reference's encoder drives: src/encoder/PageEncoder.cpp:87-147).
        uint32_t longest = 0;
        for (int m = 0; m < nc; ++m) longest = std::max(longest, cand[m].len);
        for (int si = 0; si < nstarts; ++si) {
            const uint32_t i = starts[si];
            uint32_t covered = 1;                                           // lengths 2 .. covered are priced
            for (int m = 0; m < nc; ++m) {
                const uint32_t L = cand[m].len, dist = cand[m].dist;
                if (L <= covered) continue;
                if (L >= 96) { relax_range(i, p, L, L, dist); continue; }  // long match: whole length only
                const uint32_t lmin = dist == node[i].ring[0] ? 2u : std::min(L, 4u);
                relax_range(i, p, std::max(lmin, covered + 1u), L, dist);
                covered = L;
            }
        }
        P.insert(p);
        if (longest >= 128) {                                              // skip through long matches (runs)
            const uint32_t end = p + longest;
            for (uint32_t q = p + 1; q < end && q < n; ++q) { if (node[q].cost < kInf) push_start(q); P.insert(q); }
            p = end - 1;
        }
    }
    // the tail: literals after the last command boundary go into an insert-only command
    uint32_t best_i = 0; float best = kInf;
    for (uint32_t i = 0; i <= n; ++i) {
        if (node[i].cost >= kInf) continue;
        float c = node[i].cost + (litcum[n] - litcum[i]);
        if (i < n) { const uint32_t ic = code_of(kInsBase, n - i); c += cm.icp[704 + ic] + (float)kInsExtra[ic]; }
        if (c < best) { best = c; best_i = i; }
    }
    std::vector<Command> rev;
    for (uint32_t i = best_i; i != 0; i = node[i].from) {
        Command c{}; c.insert_len = node[i].ins; c.copy_len = node[i].len; c.dist = node[i].dist;
        rev.push_back(c);
    }
    cmds.assign(rev.rbegin(), rev.rend());
    literals.clear();
    uint32_t pos = 0;
    for (auto& c : cmds) {
        literals.insert(literals.end(), data + pos, data + pos + c.insert_len);
        pos += c.insert_len + c.copy_len;
    }
    if (pos < n) {
        Command c{}; c.insert_len = n - pos; c.copy_len = 0; c.dist = 0;
        literals.insert(literals.end(), data + pos, data + n);
        cmds.push_back(c);
    }
}

void emit_command(BitWriter& w, const Command& c, const PrefixCode& icp, const PrefixCode& dist)
{
    icp.put(w, c.icp_sym);
    uint32_t ic = code_of(kInsBase, c.insert_len);
    w.put(c.insert_len - kInsBase[ic], kInsExtra[ic]);
    if (c.copy_len) {
        uint32_t cc = code_of(kCopyBase, c.copy_len);
        w.put(c.copy_len - kCopyBase[cc], kCopyExtra[cc]);
        if (c.has_dist_sym) { dist.put(w, c.dist_sym); w.put(c.dist_extra, c.dist_nbits); }
    }
}

// Returns the compressed page (empty = store raw).
std::vector<uint8_t> encode_page_once(const uint8_t* data, uint32_t n, const BrotligEncodeOptions& o, bool is_delta)
{
    std::vector<uint8_t> out;
    if (o.flags & BROTLIG_ENC_FORCE_STORED) return out;
    std::vector<Command> cmds; std::vector<uint8_t> lits;
    parse_page(data, n, o, cmds, lits);
    uint32_t npostfix = o.npostfix & 3, ndirect = (o.ndirect_m & 15) << npostfix;
    const bool use_ring = !(o.flags & BROTLIG_ENC_NO_RING_CODES);
    if (o.flags & BROTLIG_ENC_SEARCH_DIST_PARAMS) {
        // NPOSTFIX / NDIRECT search (the reference does one per page too, PageEncoder.cpp:324-377): the
        // parse is fixed, so every candidate is priced by the entropy of its distance symbols plus
        // their extra bits, and the cheapest pair is kept.
        double best = 1e300;
        uint32_t best_np = npostfix, best_nd = ndirect;
        for (uint32_t np = 0; np < 4; ++np) {
            for (uint32_t m : {0u, 1u, 2u, 4u, 8u, 12u, 15u}) {
                std::vector<Command> trial = cmds;
                assign_symbols(trial, np, m << np, use_ring);
                std::vector<uint32_t> hd(kDistAlphabet, 0);
                double bits = 0; uint32_t total = 0;
                for (auto& c : trial) if (c.has_dist_sym) { ++hd[c.dist_sym]; ++total; bits += c.dist_nbits; }
                for (uint32_t cnt : hd) if (cnt) bits += cnt * std::log2((double)total / cnt);
                uint32_t used = 0; for (uint32_t cnt : hd) used += cnt != 0;
                bits += 5.0 * used;                                    // rough price of describing the code
                if (bits < best) { best = bits; best_np = np; best_nd = m << np; }
            }
        }
        npostfix = best_np; ndirect = best_nd;
    }
    assign_symbols(cmds, npostfix, ndirect, use_ring);

    std::vector<uint32_t> hicp(kIcpAlphabet, 0), hdist(kDistAlphabet, 0), hlit(kLitAlphabet, 0);
    auto histograms = [&]() {
        std::fill(hicp.begin(), hicp.end(), 0u); std::fill(hdist.begin(), hdist.end(), 0u); std::fill(hlit.begin(), hlit.end(), 0u);
        for (auto& c : cmds) { ++hicp[c.icp_sym]; if (c.has_dist_sym) ++hdist[c.dist_sym]; }
        ++hicp[704];
        for (uint8_t b : lits) ++hlit[b];
    };
    histograms();
    if ((o.flags & BROTLIG_ENC_OPTIMAL_PARSE) && !(o.flags & BROTLIG_ENC_LITERALS_ONLY) && n >= 16) {
        // two rounds: each re-parses under the symbol costs of the previous parse
        for (int round = 0; round < 2; ++round) {
            CostModel cm;
            cm.from_histograms(hlit, hicp, hdist);
            parse_page_optimal(data, n, o, npostfix, ndirect, cm, cmds, lits);
            assign_symbols(cmds, npostfix, ndirect, use_ring);
            histograms();
        }
    }
    uint32_t pad_lit = 0;
    for (uint32_t s = 1; s < 256; ++s) if (hlit[s] > hlit[pad_lit]) pad_lit = s;   // PageEncoder.cpp:518-537

    Streams S;
    PrefixCode picp = emit_prefix_code(S, hicp, o.flags);
    PrefixCode pdist = emit_prefix_code(S, hdist, o.flags);
    PrefixCode plit = emit_prefix_code(S, hlit, o.flags);
    if (hlit[pad_lit] == 0 && plit.depth[pad_lit] == 0) {
        // empty literal alphabet -> trivial symbol 0, zero bits per literal: padding is free
        pad_lit = 0;
    }

    // rounds (src/decoder/PageDecoder.cpp:174-206 is the contract)
    size_t ci = 0, li = 0; uint32_t prev_tail = 0; bool done = false;
    while (!done) {
        S.reset();
        uint32_t nround = 0, litcount = 0;
        while (nround < kNumStreams) {
            if (ci == cmds.size()) { picp.put(S.at(), 704); done = true; break; }
            emit_command(S.at(), cmds[ci], picp, pdist);
            litcount += cmds[ci].insert_len;
            ++ci; ++nround; S.next();
        }
        S.reset();
        uint32_t ac = litcount > prev_tail ? litcount - prev_tail : 0;
        uint32_t mult = nround ? (ac + nround - 1) / nround : 0;
        uint32_t rlit = nround * mult;
        prev_tail = rlit + prev_tail - litcount;
        for (uint32_t j = 0; j < rlit; ++j) {
            uint32_t sym = li < lits.size() ? lits[li] : pad_lit;
            ++li;
            plit.put(S.at(), sym);
            S.next();
        }
    }

    // serialise: header | size table | pad | sub-streams | pad  (BrotligSwizzler.cpp:68-189)
    uint32_t len[kNumStreams], tot = 0, minlen = 0xFFFFFFFFu, maxlen = 0;
    for (uint32_t i = 0; i < kNumStreams; ++i) {
        S.w[i].flush();
        len[i] = (uint32_t)S.w[i].bytes.size();
        tot += len[i]; minlen = std::min(minlen, len[i]); maxlen = std::max(maxlen, len[i]);
    }
    const uint32_t dsb = std::max(1u, bit_width(maxlen - minlen));
    uint32_t Ssz = ((tot + 3) & ~3u) + 8, hdr_bytes = 0;
    for (int it = 0; it < 16; ++it) {
        uint32_t B = bit_width((Ssz + kNumStreams - 1) / kNumStreams), D = bit_width(bit_width(Ssz - 1));
        hdr_bytes = (((8 + B + D + kNumStreams * dsb) + 31) / 32) * 4;
        uint32_t nS = hdr_bytes + ((tot + 3) & ~3u);
        if (nS == Ssz) break;
        Ssz = nS;
    }
    if (Ssz >= n) return out;                                          // not smaller -> stored (PageEncoder.cpp:565-568)
    BitWriter h;
    // (with BROTLIG_ENC_DECODER_CORNERS: IS_DELTA set on a page of a stream that is not pre-conditioned, where the reader drops it, PageDecoder.cpp:87-88,
    // and the reserved bit set, :89)
    const bool corners = (o.flags & BROTLIG_ENC_DECODER_CORNERS) != 0;
    h.put(npostfix, 2); h.put(ndirect >> npostfix, 4); h.put((is_delta || (corners && !o.precondition)) ? 1 : 0, 1); h.put(corners ? 1 : 0, 1);
    h.put(minlen, bit_width((Ssz + kNumStreams - 1) / kNumStreams));
    h.put(dsb, bit_width(bit_width(Ssz - 1)));
    for (uint32_t i = 0; i < kNumStreams; ++i) h.put(len[i] - minlen, dsb);
    h.flush();
    out = h.bytes;
    out.resize(hdr_bytes, 0);
    for (uint32_t i = 0; i < kNumStreams; ++i) out.insert(out.end(), S.w[i].bytes.begin(), S.w[i].bytes.end());
    out.resize(Ssz, 0);
    return out;
}

// With the optimal parse asked for, the page is encoded both ways and the smaller result kept: the
// cost model prices symbols, not the description of the prefix codes, and on very compressible pages
// (a few hundred bytes) a parse that uses more distinct symbols can lose more there than it gains.
std::vector<uint8_t> encode_page(const uint8_t* data, uint32_t n, const BrotligEncodeOptions& o, bool is_delta)
{
    if (!(o.flags & BROTLIG_ENC_OPTIMAL_PARSE)) return encode_page_once(data, n, o, is_delta);
    BrotligEncodeOptions lazy = o;
    lazy.flags &= ~(uint32_t)BROTLIG_ENC_OPTIMAL_PARSE;
    std::vector<uint8_t> a = encode_page_once(data, n, lazy, is_delta), b = encode_page_once(data, n, o, is_delta);
    const size_t sa = a.empty() ? n : a.size(), sb = b.empty() ? n : b.size();   // empty = stored raw
    return sb < sa ? b : a;
}

// ---------------------------------------------------------------------------------------
// Pre-conditioning (encode side)
struct Dc {
    bool swizzle, aligned; uint32_t format, mips;
    uint32_t block_bytes, block_px, nsub, sub_size[6], sub_off[6], ncol, col[4];
    uint32_t w[33], h[33], pitch[33], nblk[33], sso[7], mob[33], mobl[33], total_blocks;
};
bool dc_setup(Dc& p, const BrotligEncodeOptions& o, uint32_t in_size)
{
    static const struct { uint32_t bytes, nsub, sizes[6], ncol, col[4]; } fmt[6] = {
        {1, 1, {1, 0, 0, 0, 0, 0}, 0, {0, 0, 0, 0}}, {8, 3, {2, 2, 4, 0, 0, 0}, 2, {0, 1, 0, 0}},
        {16, 4, {8, 2, 2, 4, 0, 0}, 2, {1, 2, 0, 0}}, {16, 6, {1, 1, 6, 2, 2, 4}, 2, {3, 4, 0, 0}},
        {8, 3, {1, 1, 6, 0, 0, 0}, 2, {0, 1, 0, 0}}, {16, 6, {1, 1, 6, 1, 1, 6}, 4, {0, 1, 3, 4}}};
    memset(&p, 0, sizeof p);
    if (o.format < 1 || o.format > 5) return false;
    p.swizzle = o.swizzle != 0; p.aligned = o.pitch_d3d12_aligned != 0; p.format = o.format;
    p.mips = o.num_mips ? o.num_mips : 1;
    p.block_bytes = fmt[o.format].bytes; p.block_px = 4; p.nsub = fmt[o.format].nsub;
    for (int i = 0; i < 6; ++i) p.sub_size[i] = fmt[o.format].sizes[i];
    p.ncol = fmt[o.format].ncol; for (int i = 0; i < 4; ++i) p.col[i] = fmt[o.format].col[i];
    p.w[0] = o.width_blocks; p.h[0] = o.height_blocks;
    if (!p.w[0] || !p.h[0] || p.w[0] > 32768 || p.h[0] > 32768 || p.mips > 32) return false;
    auto rup = [](uint32_t v, uint32_t a) { return (v + a - 1) / a * a; };
    p.pitch[0] = o.pitch_bytes ? o.pitch_bytes : (p.aligned ? rup(p.w[0] * p.block_bytes, 256) : p.w[0] * p.block_bytes);
    if (p.pitch[0] < p.w[0] * p.block_bytes || p.pitch[0] > (1u << 19)) return false;
    p.total_blocks = p.nblk[0] = p.w[0] * p.h[0];
    uint32_t mw = (p.w[0] * 4) / 2, mh = (p.h[0] * 4) / 2;
    for (uint32_t m = 1; m <= p.mips; ++m) {
        if (m < p.mips) {
            p.w[m] = (mw + 3) / 4; p.h[m] = (mh + 3) / 4; p.nblk[m] = p.w[m] * p.h[m];
            p.pitch[m] = p.aligned ? rup(p.w[m] * p.block_bytes, 256) : p.w[m] * p.block_bytes;
            p.total_blocks += p.nblk[m];
        }
        p.mob[m] = p.mob[m - 1] + p.pitch[m - 1] * p.h[m - 1];
        p.mobl[m] = p.mobl[m - 1] + p.nblk[m - 1];
        mw /= 2; mh /= 2;
    }
    if (p.mob[p.mips] != in_size) return false;
    for (uint32_t s = 1; s <= p.nsub; ++s) {
        if (s < p.nsub) p.sub_off[s] = p.sub_off[s - 1] + p.sub_size[s - 1];
        p.sso[s] = p.sso[s - 1];
        for (uint32_t m = 0; m < p.mips; ++m) p.sso[s] += p.nblk[m] * p.sub_size[s - 1];
    }
    return true;
}

// src/common/BrotligDataConditioner.cpp:28-119
void condition(const Dc& p, const uint8_t* in, uint32_t size, std::vector<uint8_t>& out)
{
    std::vector<uint8_t> temp(in, in + size);
    if (p.swizzle) {
        for (uint32_t m = 0; m < p.mips; ++m) {
            uint32_t W = p.w[m], H = p.h[m];
            if (W < 2 || H < 2) continue;
            uint8_t* base = temp.data() + p.mob[m];
            std::vector<uint8_t> src(base, base + (size_t)p.pitch[m] * H);
            uint32_t effW = W - W % 2, effH = H - H % 2, orow = 0, ocol = 0;
            for (uint32_t r = 0; r < effH; r += 2)
                for (uint32_t c = 0; c < effW; c += 2)
                    for (uint32_t ro = 0; ro < 2; ++ro)
                        for (uint32_t co = 0; co < 2; ++co) {
                            memcpy(base + (size_t)orow * p.pitch[m] + (size_t)ocol * p.block_bytes,
                                   src.data() + (size_t)(r + ro) * p.pitch[m] + (size_t)(c + co) * p.block_bytes, p.block_bytes);
                            if (++ocol == effW) { ocol = 0; ++orow; }
                        }
        }
    }
    out.assign(size, 0);
    uint32_t ptr[6];
    for (int s = 0; s < 6; ++s) ptr[s] = p.sso[s];
    for (uint32_t m = 0; m < p.mips; ++m)
        for (uint32_t r = 0; r < p.h[m]; ++r)
            for (uint32_t c = 0; c < p.w[m]; ++c) {
                uint32_t idx = p.mob[m] + r * p.pitch[m] + c * p.block_bytes;
                for (uint32_t s = 0; s < p.nsub; ++s) {
                    memcpy(&out[ptr[s]], &temp[idx], p.sub_size[s]);
                    idx += p.sub_size[s]; ptr[s] += p.sub_size[s];
                }
            }
}

// PageEncoder.cpp:576-612 (inverse of PageDecoder::DeltaDecode)
void delta_encode_page(const Dc& p, size_t page_start, size_t page_end, uint8_t* data)
{
    for (uint32_t i = 0; i < p.ncol; ++i) {
        size_t cs = p.sso[p.col[i]], ce = p.sso[p.col[i] + 1];
        if (cs < page_end && page_start < ce) {
            size_t s = cs > page_start ? cs - page_start : 0;
            size_t e = ce < page_end ? ce - page_start : page_end - page_start;
            for (size_t el = e; el-- > s + 1;) data[el] = (uint8_t)(data[el] - data[el - 1]);
        }
    }
}

}  // namespace

extern "C" uint32_t BrotligEncMaxCompressedSize(uint32_t input_size, uint32_t page_size)
{
    if (!page_size) page_size = 65536;
    uint64_t pages = ((uint64_t)input_size + page_size - 1) / page_size;
    return (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, 16 + 4 * pages + (uint64_t)input_size + 64);
}

extern "C" int BrotligEncode(uint32_t input_size, const uint8_t* src, uint32_t* output_size, uint8_t* output,
                             const BrotligEncodeOptions* opt)
{
    BrotligEncodeOptions o; memset(&o, 0, sizeof o);
    if (opt) o = *opt;
    const uint32_t page_size = o.page_size ? o.page_size : 65536;
    // The header's two-bit page size index (inc/DataStream.h:33, :73-75: 32 KiB << index) has FOUR values.  The reference's encoder stops at
    // BROTLIG_MAX_PAGE_SIZE = 128 KiB (inc/common/BrotligConstants.h:85), its decoders do not look: a stream with index 3 is 256 KiB pages to
    // DecodeCPU, so the decode path has to take it and this input generator can produce it.
    if (page_size != 32768 && page_size != 65536 && page_size != 131072 && page_size != 262144) return BROTLIG_ENC_ERROR_PAGE_SIZE;
    if (input_size == 0) return BROTLIG_ENC_ERROR_EMPTY;
    const uint64_t num_pages = ((uint64_t)input_size + page_size - 1) / page_size;
    if (num_pages > 65535) return BROTLIG_ENC_ERROR_TOO_MANY_PAGES;

    std::vector<uint8_t> cond;
    const uint8_t* data = src;
    Dc dc;
    if (o.precondition) {
        if (!dc_setup(dc, o, input_size)) return BROTLIG_ENC_ERROR_PRECON_PARAMS;
        condition(dc, src, input_size, cond);
        data = cond.data();
    }

    // pages are independent: workers pull page indices from one counter (the reference fans its
    // PageEncoderJob out the same way, src/BrotligEncoder.cpp:380-413); the result does not depend on
    // the number of workers
    std::vector<std::vector<uint8_t>> pages(num_pages);
    std::atomic<uint64_t> next{0};
    auto worker = [&]() {
        std::vector<uint8_t> work(page_size);
        for (uint64_t i = next.fetch_add(1); i < num_pages; i = next.fetch_add(1)) {
            const uint32_t off = (uint32_t)(i * page_size), n = std::min<uint32_t>(page_size, input_size - off);
            const uint8_t* pg = data + off;
            bool is_delta = false;
            if (o.precondition && o.delta) {
                memcpy(work.data(), pg, n);
                delta_encode_page(dc, off, (size_t)off + n, work.data());
                pg = work.data(); is_delta = true;
            }
            pages[i] = encode_page(pg, n, o, is_delta);
            if (pages[i].empty()) pages[i].assign(data + off, data + off + n);   // stored: conditioned, not delta (PageEncoder.cpp:321)
        }
    };
    uint32_t nthreads = o.num_threads ? o.num_threads : std::max(1u, std::thread::hardware_concurrency());
    nthreads = (uint32_t)std::min<uint64_t>(std::min(nthreads, 64u), num_pages);
    if (nthreads <= 1) worker();
    else {
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t < nthreads; ++t) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
    }

    uint64_t need = 8 + (o.precondition ? 8 : 0) + 4 * num_pages;
    for (auto& p : pages) need += p.size();
    if (need > *output_size) return BROTLIG_ENC_ERROR_OUTPUT_TOO_SMALL;

    uint8_t* w = output;
    const uint32_t last = input_size % page_size;                     // inc/DataStream.h:49-58
    w[0] = 5; w[1] = 5 ^ 0xFF; w[2] = (uint8_t)num_pages; w[3] = (uint8_t)(num_pages >> 8);
    uint32_t idx = page_size == 32768 ? 0 : page_size == 65536 ? 1 : page_size == 131072 ? 2 : 3;
    uint32_t w1 = idx | (last << 2) | ((o.precondition ? 1u : 0u) << 20);
    memcpy(w + 4, &w1, 4); w += 8;
    if (o.precondition) {                                              // inc/DataStream.h:89-98
        uint32_t h0 = (dc.swizzle ? 1u : 0u) | ((dc.aligned ? 1u : 0u) << 1) | ((dc.w[0] - 1) << 2) | ((dc.h[0] - 1) << 17);
        uint32_t h1 = dc.format | ((dc.mips - 1) << 8) | ((dc.pitch[0] - 1) << 13);
        memcpy(w, &h0, 4); memcpy(w + 4, &h1, 4); w += 8;
    }
    uint8_t* table = w; w += 4 * num_pages;
    uint32_t cur = 0;
    for (uint64_t i = 0; i < num_pages; ++i) {                         // src/BrotligEncoder.cpp:589-605
        memcpy(table + 4 * i, &cur, 4);
        memcpy(w + cur, pages[i].data(), pages[i].size());
        cur += (uint32_t)pages[i].size();
    }
    uint32_t lastsz = (uint32_t)pages[num_pages - 1].size();
    memcpy(table, &lastsz, 4);
    *output_size = (uint32_t)need;
    return BROTLIG_ENC_OK;
}
