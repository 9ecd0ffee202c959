// brotlig_split_kernels.h -- the page decode as TWO kernels (SURVEY.md 7.2 "fallback architecture"):
//
//   brotlig_entropy_kernel   per page: prefix-code tables, commands, distances (ring included) and literals -- phases K4..K8 of
//                            the reference shader (src/decoder/BrotliGCompute.hlsl:761-1347) -- and nothing else: what it
//                            decodes goes to global memory as one packed word per command {output position, literal position,
//                            distance} plus the literal bytes in consumption order.  No output window, no staging area, no
//                            piece bitmaps: 3.4 KiB of LDS per page and a small register file, so more pages in flight per CU.
//   brotlig_assemble_kernel  per page: LZ77 assembly (phase K9, BrotliGCompute.hlsl:1401-1419; PageDecoder.cpp:209-233) from
//                            those arrays, 32 commands per step and half-wave, with the window / dependency-level machinery of
//                            the fused kernel (brotlig_kernels.h) -- positions come ready-made, literal runs are fetched by
//                            their command's lane straight from the literal array.
//
// The fused kernel is bound by the latency of a wave's dependent chain times the waves a SIMD holds; each of these
// two holds less state than the fused one.  Price: the command and literal arrays are written and read once (~1-2x the
// algorithmic bytes, on a memory system that is ~7 % used).
#pragma once
#include "brotlig_kernels.h"

// waves per SIMD the two kernels are compiled for (register budget 512 / n, in steps of 8): A/B builds override them
#ifndef BROTLIG_E_WAVES
#define BROTLIG_E_WAVES 5
#endif
#ifndef BROTLIG_L_WAVES
#define BROTLIG_L_WAVES 5
#endif
// assembly kernel, loads issued a step ahead: 0 none, 1 the command words, 2 also the literal bytes (staged through LDS)
#ifndef BROTLIG_L_PREFETCH
#define BROTLIG_L_PREFETCH 2
#endif

namespace brotlig {

enum : uint32_t { kSlotReady = 1u, kSlotDelta = 2u };      // slot_hdr flags

// one command: output position of its first byte | position of its first literal in the literal array << 18 | distance << 36
// (all < 2^18: pages are at most 128 KiB).  Lengths are differences to the next entry; the array ends with a terminal
// entry {page bytes, literals consumed, 0}.
__device__ __forceinline__ uint64_t pack_cmd(uint32_t out_pos, uint32_t lit_pos, uint32_t dist)
{
    return (uint64_t)out_pos | ((uint64_t)lit_pos << 18) | ((uint64_t)dist << 36);
}

// ---- entropy kernel ----------------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) EntropyLds {       // one per 32-lane half
    // the three LUTs, then two build areas: while a table is built its own LUT and the 512 bytes behind it are scratch
    // (code-length LUT / counting-sort counters), and its code lengths live in an area that is free at that point:
    //   ICP (728 lengths): scratch lut_icp + lut_dist, lengths in lut_lit + build_tail
    //   distance (544):    scratch lut_dist + lut_lit, lengths in build_tail
    //   literal (256):     scratch lut_lit + build_tail[0..512), lengths in lit_lens
    uint16_t lut_icp[1 << kLutBitsIcp];
    uint16_t lut_dist[1 << kLutBitsDist];
    uint16_t lut_lit[1 << kLutBitsLit];
    uint8_t  build_tail[544];
    uint8_t  lit_lens[kLitAlphabet];
    uint32_t sorted_icp[(kIcpSymCap + 2) / 3];
    uint32_t sorted_dist[(kDistSymCap + 2) / 3];
    uint32_t sorted_lit[kLitAlphabet / 4];
    uint16_t limit[3][16] __attribute__((aligned(16)));
    uint32_t first_offs[3][16];
    uint32_t page_params;
    uint32_t page_stream;
    uint32_t ring[8] __attribute__((aligned(16)));
};
static_assert(kLutBitsIcp == 8 && kLutBitsDist == 8 && kLutBitsLit == 8, "build areas are laid out for three 512-byte LUTs");
static_assert(__builtin_offsetof(EntropyLds, build_tail) == 1536 && __builtin_offsetof(EntropyLds, lit_lens) == 1536 + 544, "build areas must follow the LUTs");
static_assert(512 + 544 >= kIcpAlphabet && 544 >= kDistAlphabet, "code lengths fit their build areas");

__device__ __forceinline__ uint8_t* build_lens(EntropyLds& L, uint32_t k)
{
    return k == 0u ? reinterpret_cast<uint8_t*>(L.lut_lit) : k == 1u ? L.build_tail : L.lit_lens;
}


// ---- the LDS-staged sub-stream ring (`north_star`: "LDS-staged page bitstreams") -------------
// A/B partner of BitReader for the entropy kernel (-DBROTLIG_E_RING=1): every lane keeps the next 64 bytes of its sub-stream in
// an LDS ring of four 16-byte blocks, refilled with 16-byte global loads; the bit window takes 32 bits from the ring per refill
// (one ds_read_b32).  Layout: 8-byte slot s of lane l at ((s * 64 + l) * 8) -- consecutive lanes in consecutive banks.  The
// block loaded last waits in registers (`flight`) and moves to the ring when two blocks or fewer are ahead of the read
// position, at which point the next load is issued: like BitReader's queue, a load is never waited for where it is issued.
// Same interface as BitReader, 4 KiB more LDS per wavefront.
#ifndef BROTLIG_E_RING
#define BROTLIG_E_RING 0
#endif
struct RingReader {
    const uint8_t* base;
    uint32_t limit16;       // last byte offset from base at which 16 bytes may be loaded
    uint64_t buf;
    uint32_t avail;
    uint32_t next;          // byte offset of the next 16-byte load
    uint32_t rd;            // dwords taken from the ring so far
    uint32_t wr;            // 16-byte blocks put into the ring so far
    Bytes16  flight;        // the block loaded last
    uint32_t* ring;         // LDS, this lane's column: dword d lives at ring[(d >> 1 & 7) * 128 + (d & 1)]

    __device__ __forceinline__ Bytes16 load16g(uint32_t rel) const
    {
        Bytes16 v;
        __builtin_memcpy(&v, base + (rel < limit16 ? rel : limit16), 16);
        return v;
    }
    __device__ __forceinline__ void push(Bytes16 v)
    {
        const uint32_t s = (wr & 3u) * 2u;                              // first 8-byte slot of the block
        const uint64_t lo = (uint64_t)v[0] | ((uint64_t)v[1] << 32), hi = (uint64_t)v[2] | ((uint64_t)v[3] << 32);
        *reinterpret_cast<uint64_t*>(ring + s * 128u) = lo;
        *reinterpret_cast<uint64_t*>(ring + (s + 1u) * 128u) = hi;
        ++wr;
    }
    __device__ __forceinline__ void init(const uint8_t* b, uint32_t lim, uint32_t start)
    {
        base = b; limit16 = lim >= 16u ? lim - 16u : 0u;
        const uint32_t a = start & ~3u, skip = (start & 3u) * 8u;
        rd = 0; wr = 0;
        push(load16g(a)); push(load16g(a + 16u));
        flight = load16g(a + 32u); next = a + 48u;
        buf = 0; avail = 0;
        refill();
        buf >>= skip; avail -= skip;
        if (avail < 32u) refill();
    }
    __device__ __forceinline__ void refill()
    {
        if (wr - (rd >> 2) <= 2u) { push(flight); flight = load16g(next); next += 16u; }
        const uint32_t w = ring[((rd >> 1) & 7u) * 128u + (rd & 1u)];
        ++rd;
        buf |= (uint64_t)w << avail;
        avail += 32u;
    }
    __device__ __forceinline__ void ensure(uint32_t n) { if (avail < n) refill(); }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return n >= 32u ? (uint32_t)buf : ((uint32_t)buf & ((1u << n) - 1u)); }
    __device__ __forceinline__ void consume(uint32_t n) { buf >>= n; avail -= n; }
    __device__ __forceinline__ uint32_t read(uint32_t n)
    {
        if (n == 0u) return 0u;
        ensure(n);
        const uint32_t v = peek(n);
        consume(n);
        return v;
    }
};
#if BROTLIG_E_RING
typedef RingReader EntropyReader;
#else
typedef BitReader EntropyReader;
#endif

struct __attribute__((aligned(16))) EntropyWaveLds {
    EntropyLds page[2];
    uint32_t len_code_tab[48];
#if BROTLIG_E_RING
    uint64_t ring[8 * 64];          // sub-stream rings: 64 bytes per lane
#elif defined(BROTLIG_E_PAD_LDS)
    uint64_t pad[8 * 64];           // A/B control: the same LDS footprint (hence occupancy) without the ring
#endif
};

__device__ inline void entropy_pages(EntropyWaveLds& W, const DecodeArgs& a)
{
    const uint32_t lane = wave::lane_id();
    const uint32_t sl = lane & 31u;
    EntropyLds& L = W.page[lane >> 5];
    uint16_t* const far_syms = a.far_syms + (size_t)blockIdx.x * (2u * kFarSymStride);
    const TableRef t_icp{L.lut_icp, L.sorted_icp, L.limit[0], L.first_offs[0], kIcpAlphabet, kLutBitsIcp, far_syms};
    const TableRef t_dist{L.lut_dist, L.sorted_dist, L.limit[1], L.first_offs[1], kDistAlphabet, kLutBitsDist, far_syms};
    const TableRef t_lit{L.lut_lit, L.sorted_lit, L.limit[2], L.first_offs[2], kLitAlphabet, kLutBitsLit, nullptr};

    const uint32_t resync_quarters = a.status[3];
    PageJob job = fetch_job(a, nullptr, 0u, false);
    bool live = false, finished = false, bad = false;
    EntropyReader br;
#if BROTLIG_E_RING
    br.base = a.in; br.limit16 = 0; br.buf = 0; br.avail = 64; br.next = 0; br.rd = 0; br.wr = 4; br.flight = Bytes16{0u, 0u, 0u, 0u};
    br.ring = reinterpret_cast<uint32_t*>(W.ring) + 2u * lane;
#else
    br.base = a.in; br.limit8 = 0; br.buf = 0; br.avail = 64; br.next = 0; br.queue = 0; br.queued = 64; br.flight = 0; br.zero = wave::opaque_zero();
#ifdef BROTLIG_E_PAD_LDS
    if (a.num_streams == 0xFFFFFFFFu) W.pad[lane] = 1;                 // (keeps the padding allocated)
#endif
#endif
    DistanceRing ring;
    PhaseClock<false> clk;
    uint32_t out_pos = 0;            // bytes of the page accounted for so far
    uint32_t lit_pos = 0;            // literals consumed so far (sum of insert lengths)
    uint32_t prev_tail = 0;          // literals decoded but not yet consumed: the literal array holds lit_pos + prev_tail bytes
    uint32_t cmd_count = 0;          // commands written so far

    for (;;) {
        // ---- page start (as in the fused kernel: a free half waits for a neighbour that is about to finish, so that the
        //      table builds coincide)
        {
            const uint32_t near_end = (live && (job.out_size - out_pos) * 4u < job.out_size * resync_quarters) ? 1u : 0u;
            const uint32_t other_near = wave::other_half(near_end);
            const bool want = !live && !finished && other_near == 0u;
            if (wave::any(want)) {
                bool tables_ok = true;
                uint32_t* const slot_hdr = a.slot_hdr;
                const bool start = start_pages(a, L, job, br, want, finished, sl, far_syms, tables_ok,
                                               [slot_hdr, sl](const PageJob& j) { if (sl == 0u) slot_hdr[2u * j.index + 1u] = 0u; },   // nothing for the assembly kernel (yet)
                                               clk);
                ring.reset(L, start, sl);
                if (start) {
                    out_pos = 0; lit_pos = 0; prev_tail = 0; cmd_count = 0; bad = false;
                    live = true;
                    if (!tables_ok) { bad = true; out_pos = job.out_size; }     // first round is refused (or a bare sentinel)
                }
            }
        }
        if (!wave::any(live)) break;
        const bool in_page = live;
        uint64_t* const my_cmds = a.cmds + (size_t)job.index * (a.cmd_cap + 1u);
        uint8_t* const my_lits = a.lits + (size_t)job.index * a.lit_stride;

        do {
        // -- 1. one command per lane, 2. the distance ring (stages shared with the fused kernel)
        const RingWords pushed = load_ring_pushes(L, ring);
        RoundCommands cmd = decode_round_commands(L, W.len_code_tab, t_icp, t_dist, br, live, sl, clk);
        resolve_distance_ring(L, ring, pushed, cmd, sl);
        const uint32_t sent_mask = cmd.sent_mask, n = cmd.n, ins = cmd.ins, copy = cmd.copy, dist = cmd.dist;
        const bool is_cmd = cmd.is_cmd;
        // -- 3. positions, and the checks that keep the assembly kernel inside its page
        const uint32_t tot = ins + copy;
        const uint32_t incl_tot = wave::half_scan_incl(tot);
        const uint32_t incl_ins = wave::half_scan_incl(ins);
        const uint32_t round_bytes = wave::half_bcast(incl_tot, 31);
        const uint32_t litcount = wave::half_bcast(incl_ins, 31);
        const uint32_t cmd_out = out_pos + incl_tot - tot;
        if (live && (round_bytes > job.out_size - out_pos || cmd_count + n > a.cmd_cap)) { bad = true; live = false; }
        const bool ok_cmd = is_cmd && live;
        if (ok_cmd && copy > 0u && !(dist != 0u && dist <= cmd_out + ins)) bad = true;
        if (ok_cmd) my_cmds[cmd_count + sl] = pack_cmd(cmd_out, lit_pos + incl_ins - ins, (copy > 0u && dist <= cmd_out + ins) ? dist : 0u);

        // -- 4. the round's literals (PageDecoder.cpp:196-206): literal j from sub-stream j mod 32, appended to the array
        const uint32_t ac = litcount > prev_tail ? litcount - prev_tail : 0u;
        const uint32_t mult = (live && n) ? div_small(min_u32(ac, 0x200000u) + n - 1u, n) : 0u;
        const uint32_t rlit = n * mult;
        {
            uint8_t* const dst = my_lits + (lit_pos + prev_tail);
            uint32_t j = sl;
            for (; wave::any(j + 32u < rlit); j += 64u) {
                if (j + 32u < rlit) {
                    uint32_t l0, l1;
                    br.ensure(30);
                    const uint32_t lit0 = decode_symbol<kLutBitsLit>(t_lit, br, l0);
                    br.consume(l0);
                    const uint32_t lit1 = decode_symbol<kLutBitsLit>(t_lit, br, l1);
                    br.consume(l1);
                    dst[j] = (uint8_t)lit0;
                    dst[j + 32u] = (uint8_t)lit1;
                } else if (j < rlit) {
                    uint32_t ll;
                    br.ensure(15);
                    const uint32_t lit = decode_symbol<kLutBitsLit>(t_lit, br, ll);
                    br.consume(ll);
                    dst[j] = (uint8_t)lit;
                    j -= 32u;                                               // (this lane is done: j + 64 >= rlit next time)
                }
            }
            if (j < rlit) {
                uint32_t ll;
                br.ensure(15);
                const uint32_t lit = decode_symbol<kLutBitsLit>(t_lit, br, ll);
                br.consume(ll);
                dst[j] = (uint8_t)lit;
            }
        }
        if (live) {
            prev_tail = rlit + prev_tail - litcount;
            lit_pos += litcount;
            out_pos += round_bytes;
            cmd_count += n;
        }
        if (sent_mask) live = false;
        } while (!wave::any(in_page && !live));

        // ---- page end: terminal entry and the slot header, or the error status
        const bool ended = in_page && !live;
        if (ended && out_pos != job.out_size) bad = true;
        if (ended && sl == 0u) {
            if (!bad) {
                my_cmds[cmd_count] = pack_cmd(out_pos, lit_pos, 0u);
                a.slot_hdr[2u * job.index] = cmd_count;
                a.slot_hdr[2u * job.index + 1u] = kSlotReady | ((L.page_params >> 16) != 0u ? kSlotDelta : 0u);
            } else atomicOr(a.status, kStatusBadPage);
        }
    }
}

// ---- assembly kernel ---------------------------------------------------------------------------------------------------------
constexpr uint32_t kLitDirect = 32;         // literal runs up to this length are fetched by their command's own lane

struct __attribute__((aligned(16))) AssembleLds {      // one per 32-lane half
    uint64_t stage[kStageBytes / 8];        // source bytes of far copies that do not go straight to their place
    uint32_t start_bits[kRoundMax / 32];
    uint8_t  start_cum[kRoundMax / 32];
    uint8_t  win[kWin + 16] __attribute__((aligned(16)));
#if BROTLIG_L_PREFETCH >= 2
    uint8_t  lit_stage[512 + 16] __attribute__((aligned(16)));     // the next 512 bytes of the literal array, fetched a step ahead
#endif
};
struct __attribute__((aligned(16))) AssembleWaveLds { AssembleLds page[2]; };
constexpr int kLPrefetch = BROTLIG_L_PREFETCH;

// Pages from the second work counter; for each, the entropy kernel's slot: 32 commands per step and half-wave, positions
// read from the command words, then the group loop of the fused kernel (brotlig_kernels.h, decode_pages, steps 3b-5b) with
// the literal queue replaced by loads from the literal array.  The entropy kernel has checked every length and distance.
__device__ inline void assemble_pages(AssembleWaveLds& W, const DecodeArgs& a)
{
    PhaseClock<false> clk;
    const uint32_t lane = wave::lane_id();
    const uint32_t sl = lane & 31u;
    AssembleLds& L = W.page[lane >> 5];

    PageJob job = fetch_job(a, nullptr, 0u, false);
    bool live = false, finished = false;
    uint32_t ncmd = 0, c0 = 0;       // commands of the page, first command of the next step
    uint32_t is_delta = 0;
    uint32_t out_pos = 0;
    OutView view{L.win, 0u};
    uint32_t flushed = 0;
    const uint64_t* my_cmds = a.cmds;
    const uint8_t* my_lits = a.lits;
    uint64_t pw0 = 0, pw1 = 0;       // the words of the step that starts at c0, loaded during the step before
    uint32_t staged_base = 0xFFFFFFFFu;     // literal-array position of lit_stage[0] (none staged yet)

    for (;;) {
        // ---- page start: a free half takes the next page that has a ready slot (stored and rejected pages have none)
        if (wave::any(!live && !finished)) {
            const uint32_t total = a.page_base[a.num_streams];
            const uint32_t* const order = (a.order != nullptr && total <= a.order_cap) ? a.order : nullptr;
            uint32_t* const counter = a.work_counter2;
            bool need = !live && !finished;
            while (wave::any(need)) {
                uint32_t g = 0;
                if (need && sl == 0u) g = atomicAdd(counter, 1u);
                g = wave::half_bcast(g, 0u);
                const bool got = need && g < total;
                if (need && !got) { finished = true; need = false; }
                {
                    const PageJob nj = fetch_job(a, order, g, got);
                    if (got) job = nj;
                }
                uint32_t flags = 0, n = 0;
                if (got && job.valid) { n = a.slot_hdr[2u * job.index]; flags = a.slot_hdr[2u * job.index + 1u]; }
                if (got && (flags & kSlotReady) != 0u && n != 0u) {
                    ncmd = n; c0 = 0; is_delta = (flags & kSlotDelta) != 0u ? 1u : 0u;
                    out_pos = 0; flushed = 0; view.win_base = 0u;
                    my_cmds = a.cmds + (size_t)job.index * (a.cmd_cap + 1u);
                    my_lits = a.lits + (size_t)job.index * a.lit_stride;
                    live = true; need = false;
                    staged_base = 0xFFFFFFFFu;
                    if (kLPrefetch >= 1 && sl < min_u32(32u, n)) { pw0 = my_cmds[sl]; pw1 = my_cmds[sl + 1u]; }
                }
            }
        }
        if (!wave::any(live)) break;
        const bool in_page = live;

        do {
        // -- one command per lane: its word and its successor's (the array ends with a terminal entry)
        const uint32_t n = live ? min_u32(32u, ncmd - c0) : 0u;
        const bool is_cmd = sl < n;
        uint64_t w0 = 0, w1 = 0;
        if (kLPrefetch >= 1) {
            if (is_cmd) { w0 = pw0; w1 = pw1; }
            const uint32_t cn = c0 + n;                                 // the next step's words: in flight during this one
            if (live && cn + sl < ncmd) { pw0 = my_cmds[cn + sl]; pw1 = my_cmds[cn + sl + 1u]; }
        } else if (is_cmd) { w0 = my_cmds[c0 + sl]; w1 = my_cmds[c0 + sl + 1u]; }
        const uint32_t cmd_out = (uint32_t)w0 & 0x3FFFFu, my_lit_pos = (uint32_t)(w0 >> 18) & 0x3FFFFu, dist = (uint32_t)(w0 >> 36) & 0x3FFFFu;
        const uint32_t next_out = (uint32_t)w1 & 0x3FFFFu, next_lit = (uint32_t)(w1 >> 18) & 0x3FFFFu;
        const uint32_t ins = next_lit - my_lit_pos, tot = next_out - cmd_out, copy = tot - ins;
        const uint32_t round_end = wave::half_bcast(next_out, n ? n - 1u : 0u);
        const uint32_t round_bytes = live ? round_end - out_pos : 0u;
        const uint32_t rel0 = cmd_out - out_pos;                        // my first byte, relative to the step
        // the 512 bytes of the literal array behind this step's literals: requested now, stored to LDS at the step's end
        uint64_t lp0 = 0, lp1 = 0;
        uint32_t next_stage = 0xFFFFFFFFu;
        if (kLPrefetch >= 2) {
            const uint32_t nb = wave::half_bcast(next_lit, n ? n - 1u : 0u);        // first literal of the next step
            if (live && c0 + n < ncmd && nb + 512u + 16u <= a.lit_stride) {
                next_stage = nb;
                lp0 = load_u64u(my_lits + nb + 16u * sl); lp1 = load_u64u(my_lits + nb + 16u * sl + 8u);
            }
        }
        const bool ok_cmd = is_cmd;
        const bool cp = ok_cmd && copy > 0u && dist != 0u;
        clk.lap(kPhPositions);

        // The round's output is assembled in the LDS window in byte ranges ("groups") of at most
        // kRoundMax bytes -- nearly always a single group.  A command that crosses a group boundary
        // contributes a piece to each group; a copy piece past the first is an ordinary copy from
        // `dist` bytes back (its earlier bytes are final by then).
        const uint32_t ngroups = live ? (round_bytes + kRoundMax - 1u) / kRoundMax : 0u;
        for (uint32_t g = 0; wave::any(g < ngroups); ++g) {
            const bool on = g < ngroups;
            const uint32_t g0 = g * kRoundMax, g1 = on ? min_u32(round_bytes, g0 + kRoundMax) : g0;
            const uint32_t gpos = out_pos + g0;                         // page position of the group's first byte

            // -- 3b. flush the finished bytes, slide the window when the group does not fit
            flush_and_slide(view, flushed, job.out, on, gpos, out_pos + g1, sl);
            clk.lap(kPhSlide);
            clk.count(kPhGroups, 1);
            clk.halves(kPhGroupHalves, on);
            const uint32_t span0 = gpos - view.win_base;                // window index of the group's first byte

            // -- 3c. my pieces in this group
            const uint32_t cs = rel0 + ins;                             // my copy starts here (round-relative)
            const bool in_group = on && ok_cmd && rel0 < g1 && rel0 + tot > g0;
            const uint32_t la = rel0 > g0 ? rel0 : g0, lb = cs < g1 ? cs : g1;
            const uint32_t nlit = (in_group && lb > la) ? lb - la : 0u;     // my literal bytes in the group
            const uint32_t ca = cs > g0 ? cs : g0, cb = rel0 + tot < g1 ? rel0 + tot : g1;
            const uint32_t plen = (in_group && cp && cb > ca) ? cb - ca : 0u;       // my copy bytes in the group
            const uint32_t pdst = out_pos + ca;                         // page position of the piece
            const uint32_t psrc = pdst - dist;
            const uint32_t pattern = min_u32(plen, dist);
            const uint32_t src_end = psrc + pattern;
            // the first far_len bytes of the pattern lie below the window: fetched from global memory
            // into the staging area (loads issued now, consumed after the literal decode)
            const uint32_t far_len = (plen && psrc < view.win_base && !(kAblate & kAblFar)) ? min_u32(pattern, view.win_base - psrc) : 0u;
            // A piece that lies below the window as a whole, does not overlap itself and is at most kShortCopy bytes
            // long (far_len == plen) never touches the staging area: its own lane fetches it and its bytes go from
            // these registers straight to their place in the window once the literals are decoded.  Pieces of 8 bytes
            // and more are covered by 8-byte chunks at offsets 0, 8, 16, 24 clipped to plen - 8 (the last chunk ends
            // exactly at the piece's end and overlaps its predecessor); shorter ones by one load and a split store.
            // Everything else that reaches below the window is staged: longer pieces, and patterns that straddle
            // the window boundary.  Staged pieces of up to kShortCopy bytes are fetched by their own lane too; as
            // soon as one is longer, all staged pieces get teams of lanes (two chunks per lane now, the rest later).
            const FarSources far = fetch_far_sources(job.out, nullptr, 0u, false, plen, psrc, far_len, sl);
            const bool far_direct = far.direct;
            const uint32_t stage_off = far.stage_off;
            clk.lap(kPhPieces);
            // my literal run in this group: fetched from the page's literal array by my own lane, now; stored to the window
            // once the dependency analysis below has covered the latency.  Runs of 8 bytes and more as 8-byte chunks at
            // offsets clipped to nlit - 8; shorter ones as one load and a split store (the array has 64 bytes of slack).
            const uint8_t* const lsrc = my_lits + (my_lit_pos + (la - rel0));
            const uint32_t lclip = nlit >= 8u ? nlit - 8u : 0u;
            uint64_t le0 = 0, le1 = 0, le2 = 0, le3 = 0;
            const uint32_t lfirst = my_lit_pos + (la - rel0);
            bool staged_lit = false;                                    // my run lies in the part of the array staged in LDS
#if BROTLIG_L_PREFETCH >= 2
            staged_lit = nlit != 0u && nlit <= kLitDirect && lfirst >= staged_base && lfirst + nlit <= staged_base + 512u;
            if (staged_lit) {
                const uint8_t* q = L.lit_stage + (lfirst - staged_base);
                le0 = load_u64u(q);
                if (nlit > 8u) le1 = load_u64u(q + min_u32(8u, lclip));
                if (nlit > 16u) le2 = load_u64u(q + min_u32(16u, lclip));
                if (nlit > 24u) le3 = load_u64u(q + lclip);
            }
#endif
            if (nlit != 0u && nlit <= kLitDirect && !staged_lit) {
                le0 = load_u64u(lsrc);
                if (nlit > 8u) le1 = load_u64u(lsrc + min_u32(8u, lclip));
                if (nlit > 16u) le2 = load_u64u(lsrc + min_u32(16u, lclip));
                if (nlit > 24u) le3 = load_u64u(lsrc + lclip);
            }
            const uint32_t dep_mask = piece_dependencies<PhaseClock<false>>(L.start_bits, L.start_cum, on, wave::ballot64(in_group), (rel0 > g0 ? rel0 : g0) - g0, gpos,
                                                         psrc, src_end, plen != 0u && !(kAblate & kAblDeps), sl, clk);
            clk.lap(kPhCopyFence);

            // -- 4. literal runs to their place in the window (PageDecoder.cpp:209-211)
            {
                uint8_t* const ld = L.win + (span0 - g0 + la);
                if (nlit != 0u && nlit <= kLitDirect) {
                    if (nlit >= 8u) {
                        __builtin_memcpy(ld, &le0, 8);
                        if (nlit > 8u) __builtin_memcpy(ld + min_u32(8u, lclip), &le1, 8);
                        if (nlit > 16u) __builtin_memcpy(ld + min_u32(16u, lclip), &le2, 8);
                        if (nlit > 24u) __builtin_memcpy(ld + lclip, &le3, 8);
                    } else store_bytes(ld, le0, nlit);
                }
                if (wave::any(nlit > kLitDirect)) {                     // long inserts: teams of lanes, global memory -> window
                    const uint32_t lmask = wave::half_ballot(nlit > kLitDirect);
                    const Team tl = make_team(lmask, sl);
                    const uint32_t l_src = wave::half_shfl(my_lit_pos + (la - rel0), tl.job), l_dst = wave::half_shfl(span0 - g0 + la, tl.job);
                    const uint32_t l_len = wave::half_shfl(nlit, tl.job);
                    const bool act = tl.serves && lmask != 0u;
                    for (uint32_t c = tl.member; wave::any(act && 8u * c < l_len); c += 1u << tl.log2_size) {
                        const uint32_t j = 8u * c;
                        if (act && j < l_len) store_bytes(L.win + l_dst + j, load_u64u(my_lits + l_src + j), l_len - j);
                    }
                }
            }
            wave::sync();
            clk.lap(kPhLiterals);

            // -- 5a. far sources: short whole pieces straight into the window, everything else into the
            //        staging area (aligned 8-byte LDS writes)
            const uint32_t src_idx = psrc - view.win_base;              // window index of the pattern start (negative when far)
            const uint32_t dst_idx = pdst - view.win_base;
            store_far_sources(L.win, L.stage, job.out, far, plen, far_len, dst_idx);
            wave::sync();
            clk.lap(kPhLvLong);

            // -- 5b. LZ77 copies in dependency levels
            copy_levels(L.win, L.stage, plen, dist, far_len, stage_off, src_idx, dst_idx, far.direct_w, dep_mask, sl, false, clk);
            clk.lap(kPhCopyLevels);
        }


#if BROTLIG_L_PREFETCH >= 2
        wave::sync();
        if (next_stage != 0xFFFFFFFFu) { uint64_t v[2] = {lp0, lp1}; __builtin_memcpy(L.lit_stage + 16u * sl, v, 16); }
        if (live) staged_base = next_stage;
        wave::sync();
#endif
        if (live) { out_pos += round_bytes; c0 += n; if (c0 >= ncmd) live = false; }
        } while (!wave::any(in_page && !live));

        // ---- page end for the halves whose page is complete
        const bool ended = in_page && !live;
        wave::sync();
        if (ended) flushed = flush_window(job.out, view, flushed, out_pos, true, sl);

        delta_decode_page(job, ended && is_delta != 0u, sl);       // colour sub-streams (PageDecoder.cpp:446-471)
    }
}


// ---- assembly kernel, second form: the page is assembled IN PLACE in global memory --------------------------------------------
// No LDS window: one wavefront per page, 64 commands per step; every command's lane copies its literal run from the literal
// array and its LZ77 copy from `dist` bytes back straight into the output, in 8-byte chunks.  What makes that legal: a wave's
// vector-memory requests are performed in order, so a load issued after a store of the same wave -- any lane -- sees it (the
// fused kernel relies on the same for its far copies); copies that read what an earlier command of the same step writes run
// in dependency levels, a copy that overlaps itself replays its pattern (the first `dist` bytes, complete before it starts)
// instead of reading its own output.  The kernel keeps 260 bytes of LDS and a small register file: 32 waves per CU, and no
// flush / slide / staging / piece bitmaps at all.  Recently written lines come back from the CU's L1 and the XCD's L2; far
// sources cost what they cost the fused kernel.
#ifndef BROTLIG_G_WAVES
#define BROTLIG_G_WAVES 8
#endif
constexpr uint32_t kCoopLen = 64;          // pieces longer than this are copied by the whole wave, 8 bytes per lane and pass

struct GlobalAsmLds { uint32_t pos[68]; };      // output position of each of the step's commands, then the step's end

// `len` bytes from src to dst by the whole wave (all lanes call it with the same arguments); d = distance for a copy that may
// overlap itself (d >= len: plain), or 0xFFFFFFFF for literals.  Source bytes are all in place before the call.
__device__ __forceinline__ void coop_copy(uint8_t* dst, const uint8_t* src, uint32_t len, uint32_t d, uint32_t lane)
{
    for (uint32_t j = 8u * lane; j < len; j += 512u) {
        uint64_t v;
        if (d >= len) v = load_u64u(src + j);
        else v = pattern_source8(src, d, mod_u16(j, d));
        store_bytes(dst + j, v, len - j);
    }
}

__device__ inline void assemble_pages_global(GlobalAsmLds& L, const DecodeArgs& a)
{
    const uint32_t lane = wave::lane_id();
    const uint32_t total = a.page_base[a.num_streams];
    const uint32_t* const order = (a.order != nullptr && total <= a.order_cap) ? a.order : nullptr;
    for (;;) {
        uint32_t g = 0;
        if (lane == 0u) g = atomicAdd(a.work_counter2, 1u);
        g = wave::uniform(wave::bcast(g, 0u));
        if (g >= total) break;
        const PageJob job = fetch_job(a, order, g, true);
        uint32_t ncmd = 0, flags = 0;
        if (job.valid) { ncmd = a.slot_hdr[2u * job.index]; flags = a.slot_hdr[2u * job.index + 1u]; }
        if ((flags & kSlotReady) == 0u || ncmd == 0u) continue;
        const uint64_t* const cmds = a.cmds + (size_t)job.index * (a.cmd_cap + 1u);
        const uint8_t* const lits = a.lits + (size_t)job.index * a.lit_stride;
        uint8_t* const out = job.out;
        uint64_t pw0 = 0, pw1 = 0;
        if (lane < ncmd) { pw0 = cmds[lane]; pw1 = cmds[lane + 1u]; }

        for (uint32_t c0 = 0; c0 < ncmd; c0 += 64u) {
            const uint32_t n = min_u32(64u, ncmd - c0);
            const bool is_cmd = lane < n;
            const uint64_t w0 = pw0, w1 = pw1;
            if (c0 + 64u + lane < ncmd) { pw0 = cmds[c0 + 64u + lane]; pw1 = cmds[c0 + 65u + lane]; }     // next step's, in flight meanwhile
            const uint32_t cmd_out = (uint32_t)w0 & 0x3FFFFu, lit_pos = (uint32_t)(w0 >> 18) & 0x3FFFFu, dist = (uint32_t)(w0 >> 36) & 0x3FFFFu;
            const uint32_t next_out = (uint32_t)w1 & 0x3FFFFu, next_lit = (uint32_t)(w1 >> 18) & 0x3FFFFu;
            const uint32_t ins = is_cmd ? next_lit - lit_pos : 0u;
            const uint32_t copy = (is_cmd && dist != 0u) ? (next_out - cmd_out) - ins : 0u;
            const uint32_t cdst = cmd_out + ins, src = cdst - dist;
            const uint32_t pattern = min_u32(copy, dist), src_end = src + pattern;
            const uint32_t S0 = wave::uniform(wave::bcast(cmd_out, 0u));          // first byte of the step
            // positions of the step's commands, for the dependency search below
            wave::sync();
            if (is_cmd) L.pos[lane] = cmd_out;
            if (lane == n - 1u) L.pos[n] = next_out;
            wave::sync();

            // -- copies whose source lies below the step (nearly all far ones, most near ones): loads first, they take longest
            const bool has_copy = copy != 0u;
            const bool simple = dist >= copy;                                     // no overlap with itself
            const bool own = has_copy && copy <= kCoopLen;                         // copied by its own lane
            const bool early = own && simple && src_end <= S0;
            const uint32_t cclip = copy >= 8u ? copy - 8u : 0u;
            uint64_t e0 = 0, e1 = 0;
            if (early) {
                e0 = load_u64u(out + src);
                if (copy > 8u) e1 = load_u64u(out + src + min_u32(8u, cclip));
            }
            // -- literal runs (PageDecoder.cpp:209-211): short ones by their own lane, long ones by the whole wave
            {
                const uint32_t lclip = ins >= 8u ? ins - 8u : 0u;
                const bool lown = ins != 0u && ins <= kCoopLen;
                uint64_t l0 = 0, l1 = 0;
                if (lown) {
                    l0 = load_u64u(lits + lit_pos);
                    if (ins > 8u) l1 = load_u64u(lits + lit_pos + min_u32(8u, lclip));
                }
                if (lown) {
                    if (ins >= 8u) {
                        __builtin_memcpy(out + cmd_out, &l0, 8);
                        if (ins > 8u) __builtin_memcpy(out + cmd_out + min_u32(8u, lclip), &l1, 8);
                    } else store_bytes(out + cmd_out, l0, ins);
                }
                for (uint32_t o = 16u; wave::any(lown && ins > o); o += 16u) {       // runs of 17..64 bytes: further chunk pairs
                    if (lown && ins > o) {
                        const uint32_t c0o = min_u32(o, lclip), c1o = min_u32(o + 8u, lclip);
                        const uint64_t v0 = load_u64u(lits + lit_pos + c0o), v1 = load_u64u(lits + lit_pos + c1o);
                        __builtin_memcpy(out + cmd_out + c0o, &v0, 8);
                        __builtin_memcpy(out + cmd_out + c1o, &v1, 8);
                    }
                }
                uint64_t lmask = wave::ballot64(ins > kCoopLen);
                while (lmask != 0ull) {
                    const uint32_t k = (uint32_t)__builtin_ctzll(lmask);
                    lmask &= lmask - 1ull;
                    const uint32_t k_dst = wave::uniform(wave::bcast(cmd_out, k)), k_src = wave::uniform(wave::bcast(lit_pos, k));
                    const uint32_t k_len = wave::uniform(wave::bcast(ins, k));
                    coop_copy(out + k_dst, lits + k_src, k_len, 0xFFFFFFFFu, lane);
                }
            }
            // early copies: their bytes are here by now
            if (early) {
                if (copy >= 8u) {
                    __builtin_memcpy(out + cdst, &e0, 8);
                    if (copy > 8u) __builtin_memcpy(out + cdst + min_u32(8u, cclip), &e1, 8);
                } else store_bytes(out + cdst, e0, copy);
            }
            for (uint32_t o = 16u; wave::any(early && copy > o); o += 16u) {
                if (early && copy > o) {
                    const uint32_t c0o = min_u32(o, cclip), c1o = min_u32(o + 8u, cclip);
                    const uint64_t v0 = load_u64u(out + src + c0o), v1 = load_u64u(out + src + c1o);
                    __builtin_memcpy(out + cdst + c0o, &v0, 8);
                    __builtin_memcpy(out + cdst + c1o, &v1, 8);
                }
            }

            // -- the rest in dependency levels.  A copy waits for the commands of this step (before it) whose COPY bytes its
            //    source touches; their literals are in place already.  [lo_l, hi_l) = those commands: binary search in pos[].
            uint32_t lo_l = 0, hi_l = 0;
            const bool late = has_copy && !early;
            if (wave::any(late && src_end > S0)) {
                // largest l with pos[l] <= x, for x = max(src, S0) and x = src_end - 1 (both >= S0 = pos[0])
                const uint32_t xa = src > S0 ? src : S0, xb = src_end - 1u;
                uint32_t la = 0, lb = 0;
                for (uint32_t step = 32u; step != 0u; step >>= 1) {
                    const uint32_t ta = la + step, tb = lb + step;
                    const uint32_t pa = L.pos[min_u32(ta, n)], pb = L.pos[min_u32(tb, n)];
                    if (ta < n && pa <= xa) la = ta;
                    if (tb < n && pb <= xb) lb = tb;
                }
                if (late && src_end > S0) { lo_l = la; hi_l = min_u32(lb + 1u, lane); }
            }
            uint64_t todo = wave::ballot64(late);
            while (todo != 0ull) {
                const uint64_t window = hi_l > lo_l ? ((todo >> lo_l) & ((hi_l - lo_l) >= 64u ? ~0ull : ((1ull << (hi_l - lo_l)) - 1ull))) : 0ull;
                const bool ready = late && ((todo >> lane) & 1ull) != 0ull && window == 0ull;
                // own-lane pieces: plain ones in chunk pairs at clipped offsets, self-overlapping ones from their pattern
                const bool r_own = ready && own;
                if (r_own && simple) {
                    uint64_t v0 = load_u64u(out + src), v1 = 0;
                    if (copy > 8u) v1 = load_u64u(out + src + min_u32(8u, cclip));
                    if (copy >= 8u) {
                        __builtin_memcpy(out + cdst, &v0, 8);
                        if (copy > 8u) __builtin_memcpy(out + cdst + min_u32(8u, cclip), &v1, 8);
                    } else store_bytes(out + cdst, v0, copy);
                }
                for (uint32_t o = 16u; wave::any(r_own && simple && copy > o); o += 16u) {
                    if (r_own && simple && copy > o) {
                        const uint32_t c0o = min_u32(o, cclip), c1o = min_u32(o + 8u, cclip);
                        const uint64_t v0 = load_u64u(out + src + c0o), v1 = load_u64u(out + src + c1o);
                        __builtin_memcpy(out + cdst + c0o, &v0, 8);
                        __builtin_memcpy(out + cdst + c1o, &v1, 8);
                    }
                }
                for (uint32_t o = 0u; wave::any(r_own && !simple && o < copy); o += 8u) {
                    if (r_own && !simple && o < copy) store_bytes(out + cdst + o, pattern_source8(out + src, dist, mod_u16(o, dist)), copy - o);
                }
                // long pieces that are ready: the whole wave, one piece after the other
                uint64_t big = wave::ballot64(ready && !own);
                while (big != 0ull) {
                    const uint32_t k = (uint32_t)__builtin_ctzll(big);
                    big &= big - 1ull;
                    const uint32_t k_dst = wave::uniform(wave::bcast(cdst, k)), k_src = wave::uniform(wave::bcast(src, k));
                    const uint32_t k_len = wave::uniform(wave::bcast(copy, k)), k_d = wave::uniform(wave::bcast(dist, k));
                    coop_copy(out + k_dst, out + k_src, k_len, k_d, lane);
                }
                todo &= ~wave::ballot64(ready);
            }
        }

        // per-page delta decode (PageDecoder.cpp:446-471): the stage works per half-wave; this page is the lower half's
        delta_decode_page(job, (flags & kSlotDelta) != 0u && lane < 32u, lane & 31u);
    }
}


// ---- assembly kernel, third form: the whole page in LDS, one workgroup per page -----------------------------------------------
// (round 3, last experiment: what would an assembly cost that never reads its own output back from memory?)
// A workgroup of kPageWaves wavefronts owns one page of at most 64 KiB and keeps all of it in LDS while it is built, so there
// are no far copies at all: the 30 GB of back-reference line reads of the fused kernel (profiles/r03_traffic_calibration.md) do
// not exist here, and the output is written once, coalesced.  Two sweeps over the page's command array:
//   A. literal runs: literal array -> page buffer (independent of everything else);
//   B. copies, kPageThreads commands per step, one per thread, as DATAFLOW: a copy whose source lies below the step's first
//      byte runs at once; one that reads bytes of earlier commands of the same step finds those commands by binary search in
//      the step's position array and polls their done bits (LDS, workgroup scope) -- no barrier between dependency levels,
//      the wavefronts run ahead as far as their own dependencies allow.  Dependencies point to lower command indices only,
//      so the lowest unfinished command can always run.
// 64 KiB + 1.2 KiB of LDS per workgroup: two workgroups per CU.
#ifndef BROTLIG_PAGE_WAVES
#define BROTLIG_PAGE_WAVES 4
#endif
constexpr uint32_t kPageWaves = BROTLIG_PAGE_WAVES, kPageThreads = 64u * kPageWaves;
constexpr uint32_t kPageBytes = 65536;

struct __attribute__((aligned(16))) PageAsmLds {
    uint8_t  page[kPageBytes + 16];
    uint32_t pos[kPageThreads + 4];         // output position of each of the step's commands, then the step's end
    uint32_t done[kPageThreads / 32];       // bit t: command t of the step is complete
    uint32_t ctl[4];                        // [0] the page taken from the work counter
};

// every command of [lo, hi) of the step complete?
__device__ __forceinline__ bool page_deps_done(const uint32_t* done, uint32_t lo, uint32_t hi)
{
    bool ok = true;
    for (uint32_t w = lo >> 5; ok && w <= ((hi - 1u) >> 5); ++w) {
        const uint32_t first = w == (lo >> 5) ? (lo & 31u) : 0u, last = w == ((hi - 1u) >> 5) ? ((hi - 1u) & 31u) : 31u;
        const uint32_t need = (0xFFFFFFFFu >> (31u - last)) & (0xFFFFFFFFu << first);
        ok = (wave::lds_load_acquire(done + w) & need) == need;
    }
    return ok;
}

__device__ inline void assemble_pages_in_lds(PageAsmLds& L, const DecodeArgs& a)
{
    const uint32_t tid = threadIdx.x, lane = wave::lane_id();
    const uint32_t total = a.page_base[a.num_streams];
    const uint32_t* const order = (a.order != nullptr && total <= a.order_cap) ? a.order : nullptr;
    for (;;) {
        __syncthreads();
        if (tid == 0u) L.ctl[0] = atomicAdd(a.work_counter2, 1u);
        __syncthreads();
        const uint32_t g = L.ctl[0];
        if (g >= total) break;
        const PageJob job = fetch_job(a, order, g, true);
        uint32_t ncmd = 0, flags = 0;
        if (job.valid) { ncmd = a.slot_hdr[2u * job.index]; flags = a.slot_hdr[2u * job.index + 1u]; }
        if ((flags & kSlotReady) == 0u || ncmd == 0u) continue;
        if (job.out_size > kPageBytes) { if (tid == 0u) atomicOr(a.status, kStatusBadPage); continue; }      // (experiment: 64 KiB pages at most)
        const uint64_t* const cmds = a.cmds + (size_t)job.index * (a.cmd_cap + 1u);
        const uint8_t* const lits = a.lits + (size_t)job.index * a.lit_stride;

        // ---- A. literal runs (PageDecoder.cpp:209-211).  The words of the next pass are in flight while this one works.
        {
            uint64_t pw0 = 0, pw1 = 0;
            if (tid < ncmd) { pw0 = cmds[tid]; pw1 = cmds[tid + 1u]; }
            for (uint32_t c0 = 0; c0 < ncmd; c0 += kPageThreads) {
                const uint64_t w0 = pw0, w1 = pw1;
                const bool is_cmd = c0 + tid < ncmd;
                if (c0 + kPageThreads + tid < ncmd) { pw0 = cmds[c0 + kPageThreads + tid]; pw1 = cmds[c0 + kPageThreads + tid + 1u]; }
                const uint32_t cmd_out = (uint32_t)w0 & 0x3FFFFu, lit_pos = (uint32_t)(w0 >> 18) & 0x3FFFFu;
                const uint32_t ins = is_cmd ? ((uint32_t)(w1 >> 18) & 0x3FFFFu) - lit_pos : 0u;
                // runs of up to 64 bytes by their own thread, longer ones by the wavefront of their thread
                if (ins != 0u && ins <= kCoopLen)
                    for (uint32_t o = 0; o < ins; o += 8u) store_bytes(L.page + cmd_out + o, load_u64u(lits + lit_pos + o), ins - o);
                uint64_t lmask = wave::ballot64(ins > kCoopLen);
                while (lmask != 0ull) {
                    const uint32_t k = (uint32_t)__builtin_ctzll(lmask);
                    lmask &= lmask - 1ull;
                    const uint32_t k_dst = wave::uniform(wave::bcast(cmd_out, k)), k_src = wave::uniform(wave::bcast(lit_pos, k));
                    const uint32_t k_len = wave::uniform(wave::bcast(ins, k));
                    coop_copy(L.page + k_dst, lits + k_src, k_len, 0xFFFFFFFFu, lane);
                }
            }
        }
        __syncthreads();

        // ---- B. copies (PageDecoder.cpp:219-232), one step of kPageThreads commands at a time
        {
            uint64_t pw0 = 0, pw1 = 0;
            if (tid < ncmd) { pw0 = cmds[tid]; pw1 = cmds[tid + 1u]; }
            for (uint32_t c0 = 0; c0 < ncmd; c0 += kPageThreads) {
                const uint32_t n = min_u32(kPageThreads, ncmd - c0);
                const bool is_cmd = tid < n;
                const uint64_t w0 = pw0, w1 = pw1;
                if (c0 + kPageThreads + tid < ncmd) { pw0 = cmds[c0 + kPageThreads + tid]; pw1 = cmds[c0 + kPageThreads + tid + 1u]; }
                const uint32_t cmd_out = (uint32_t)w0 & 0x3FFFFu, lit_pos = (uint32_t)(w0 >> 18) & 0x3FFFFu, dist = (uint32_t)(w0 >> 36) & 0x3FFFFu;
                const uint32_t next_out = (uint32_t)w1 & 0x3FFFFu, next_lit = (uint32_t)(w1 >> 18) & 0x3FFFFu;
                const uint32_t ins = is_cmd ? next_lit - lit_pos : 0u;
                const uint32_t copy = (is_cmd && dist != 0u) ? (next_out - cmd_out) - ins : 0u;
                const uint32_t cdst = cmd_out + ins, src = cdst - dist;
                const uint32_t src_end = src + min_u32(copy, dist);
                if (is_cmd) L.pos[tid] = cmd_out;
                if (tid == n - 1u) L.pos[n] = next_out;
                // a command without a copy is complete (its literals are in place): its bit starts set
                {
                    const uint64_t nocopy = wave::ballot64(!is_cmd || copy == 0u);
                    if (lane == 0u) { L.done[2u * (tid >> 6)] = (uint32_t)nocopy; L.done[2u * (tid >> 6) + 1u] = (uint32_t)(nocopy >> 32); }
                }
                __syncthreads();
                const uint32_t S0 = L.pos[0];
                // the commands of this step (before mine) whose bytes my source touches: [lo, hi)
                uint32_t lo = 0, hi = 0;
                if (copy != 0u && src_end > S0) {
                    const uint32_t xa = src > S0 ? src : S0, xb = src_end - 1u;
                    uint32_t la = 0, lb = 0;
                    for (uint32_t step = kPageThreads >> 1; step != 0u; step >>= 1) {
                        const uint32_t ta = la + step, tb = lb + step;
                        if (ta < n && L.pos[ta] <= xa) la = ta;
                        if (tb < n && L.pos[tb] <= xb) lb = tb;
                    }
                    lo = la; hi = min_u32(lb + 1u, tid);
                }
                bool mine = copy != 0u;
                const bool simple = dist >= copy, own = copy <= kCoopLen;
                while (wave::any(mine)) {
                    const bool ready = mine && (hi <= lo || page_deps_done(L.done, lo, hi));
                    if (ready && own) {
                        if (simple) {
                            const uint32_t clip = copy >= 8u ? copy - 8u : 0u;
                            if (copy >= 8u) for (uint32_t o = 0; o < copy; o += 8u) { const uint32_t oc = min_u32(o, clip); const uint64_t v = load_u64u(L.page + src + oc); __builtin_memcpy(L.page + cdst + oc, &v, 8); }
                            else store_bytes(L.page + cdst, load_u64u(L.page + src), copy);
                        } else {
                            for (uint32_t o = 0; o < copy; o += 8u) store_bytes(L.page + cdst + o, pattern_source8(L.page + src, dist, mod_u16(o, dist)), copy - o);
                        }
                    }
                    uint64_t big = wave::ballot64(ready && !own);       // long copies: the whole wavefront, one after the other
                    while (big != 0ull) {
                        const uint32_t k = (uint32_t)__builtin_ctzll(big);
                        big &= big - 1ull;
                        const uint32_t k_dst = wave::uniform(wave::bcast(cdst, k)), k_src = wave::uniform(wave::bcast(src, k));
                        const uint32_t k_len = wave::uniform(wave::bcast(copy, k)), k_d = wave::uniform(wave::bcast(dist, k));
                        coop_copy(L.page + k_dst, L.page + k_src, k_len, k_d, lane);
                    }
                    const bool progressed = wave::any(ready);
                    wave::global_fence();                               // the bytes before the bits (workgroup scope)
                    if (ready) { atomicOr(&L.done[tid >> 5], 1u << (tid & 31u)); mine = false; }
                    if (!progressed) wave::nap();
                }
                __syncthreads();
            }
        }
        // ---- C. the page, once, coalesced: 16 bytes per thread and pass
        {
            const uint32_t whole = job.out_size & ~15u;
            for (uint32_t o = 16u * tid; o < whole; o += 16u * kPageThreads) store16(job.out + o, load16(L.page + o));
            for (uint32_t o = whole + tid; o < job.out_size; o += kPageThreads) job.out[o] = L.page[o];
        }
        // per-page delta decode (PageDecoder.cpp:446-471): in global memory by the first half-wave, as elsewhere (could be done in LDS)
        __syncthreads();
        if (tid < 64u) delta_decode_page(job, (flags & kSlotDelta) != 0u && lane < 32u, lane & 31u);
    }
}

__global__ void __launch_bounds__(64, BROTLIG_E_WAVES) brotlig_entropy_kernel(DecodeArgs a)
{
    __shared__ EntropyWaveLds W;
    const uint32_t lane = wave::lane_id();
    if (lane < 48u) W.len_code_tab[lane] = kLenCodeTab[lane];
    wave::sync();
    entropy_pages(W, a);
}

__global__ void __launch_bounds__(64, BROTLIG_L_WAVES) brotlig_assemble_kernel(DecodeArgs a)
{
    __shared__ AssembleWaveLds W;
    assemble_pages(W, a);
}

__global__ void __launch_bounds__(64 * BROTLIG_PAGE_WAVES) brotlig_assemble_page_kernel(DecodeArgs a)
{
    __shared__ PageAsmLds L;
    assemble_pages_in_lds(L, a);
}

__global__ void __launch_bounds__(64, BROTLIG_G_WAVES) brotlig_assemble_global_kernel(DecodeArgs a)
{
    __shared__ GlobalAsmLds L;
    assemble_pages_global(L, a);
}

}  // namespace brotlig
