// brotlig_round.h -- page start, the commands and distance ring of a round, and the persistent page loop of one wavefront (PageDecoder.cpp:65-268).
// Part of the gfx950 Brotli-G decode kernels; brotlig_kernels.h includes the parts in order and says what the whole replaces.
#pragma once
#include "brotlig_tables.h"
#include "brotlig_jobs.h"
#include "brotlig_copy_levels.h"

namespace brotlig {

// ===========================================================================================
// Stages of a page decode shared by the fused kernel (decode_pages) and the entropy kernel of the
// split experiment (profiles/experiments/split_path/brotlig_split_kernels.h).  `Lds` is the per-half LDS record: both kinds carry
// lut_icp / lut_dist / lut_lit, sorted_*, limit, first_offs, page_params and ring_push under
// these names; where a table's code lengths live while it is built differs (build_lens).
// All of them run in wave-uniform control flow, with per-half predicates as operands.

template <class Lds>
__device__ __forceinline__ TableRef table_of(Lds& L, uint32_t k, uint16_t* far_syms)
{
    return TableRef{k == 0u ? L.lut_icp : k == 1u ? L.lut_dist : L.lut_lit,
                    k == 0u ? L.sorted_icp : k == 1u ? L.sorted_dist : L.sorted_lit,
                    L.limit[k], L.first_offs[k],
                    k == 0u ? kIcpAlphabet : k == 1u ? kDistAlphabet : kLitAlphabet,
                    k == 0u ? kLutBitsIcp : k == 1u ? kLutBitsDist : kLutBitsLit, far_syms};
}
// fused kernel: the output window holds the code lengths of whichever table is being built
template <class G> __device__ __forceinline__ uint8_t* build_lens(PageLdsT<G>& L, uint32_t) { return L.win; }

// ---- stage: page start.  The halves with `want` take pages from the work counter until each holds a compressed one
// (stored pages, PageDecoder.cpp:70-76, are copied on the spot; rejected ones skipped), then read the page header and
// the sub-stream size table (:79-121), start their bit readers and build the three prefix-code tables (:125-147).
// `on_pull(job)` is called by every lane of a half for every page the half takes.  Returns whether this half starts a
// page; `tables_ok` = all three descriptions were defined.
template <class Lds, class Reader, class OnPull, class Clock>
__device__ __forceinline__ bool start_pages(const DecodeArgs& a, Lds& L, PageJob& job, Reader& br, bool want, bool& finished,
                                            uint32_t sl, uint16_t* far_syms, bool& tables_ok, OnPull on_pull, Clock& clk)
{
    const uint32_t total = a.page_base[a.num_streams];
    const JobRecord* const jobs = (a.jobs != nullptr && total <= a.jobs_cap) ? a.jobs : nullptr;
    uint32_t* const work_counter = a.work_counter;
    bool need = want, start = false;
    while (wave::any(need)) {
        uint32_t g = 0;
        if (need && sl == 0u) g = atomicAdd(work_counter, 1u);
        g = wave::half_bcast(g, 0u);
        const bool got = need && g < total;
        if (need && !got) { finished = true; need = false; }
        {
            const PageJob nj = fetch_job(a, jobs, g, got);
            if (got) job = nj;
        }
        if (got) on_pull(job);
        const bool fresh = got && job.valid;
        const bool stored = fresh && job.in_size == job.out_size;
        if (stored) {                                       // plain copy: 16 bytes per lane, four loads in flight per step (round 5; 4 bytes per step
                                                            // until then -- 0.06 ms for a page whose neighbour half waits for it)
            const uint32_t vecs = job.out_size >> 4;        // (the page's output is 16-byte aligned; its input lies where the page table says)
            for (uint32_t i = sl; i < vecs; i += 128u) {
                Bytes16 v0, v1, v2, v3;
                __builtin_memcpy(&v0, job.in + 16u * i, 16);
                if (i + 32u < vecs) __builtin_memcpy(&v1, job.in + 16u * (i + 32u), 16);
                if (i + 64u < vecs) __builtin_memcpy(&v2, job.in + 16u * (i + 64u), 16);
                if (i + 96u < vecs) __builtin_memcpy(&v3, job.in + 16u * (i + 96u), 16);
                store16(job.out + 16u * i, v0);
                if (i + 32u < vecs) store16(job.out + 16u * (i + 32u), v1);
                if (i + 64u < vecs) store16(job.out + 16u * (i + 64u), v2);
                if (i + 96u < vecs) store16(job.out + 16u * (i + 96u), v3);
            }
            for (uint32_t i = (vecs << 4) + sl; i < job.out_size; i += 32u) job.out[i] = job.in[i];
        }
        if (fresh && !stored) { start = true; need = false; }
    }
    {
        uint32_t my_len = 0, hdr_bytes = 0;
        if (start) {
            const uint32_t w0 = br_load(job, 0u), w1 = br_load(job, 4u);
            const uint64_t h = (uint64_t)w0 | ((uint64_t)w1 << 32);
            const uint32_t npostfix = (uint32_t)h & 3u;
            const uint32_t is_delta = ((((uint32_t)h >> 6) & 1u) != 0u && job.dc != nullptr) ? 1u : 0u;   // PageDecoder.cpp:87-88
            // kept in LDS rather than in a register for the whole page: read once per round at most
            if (sl == 0u) { L.page_params = npostfix | ((((uint32_t)h >> 2) & 15u) << (npostfix + 8u)) | (is_delta << 16); L.page_stream = job.stream; }
            const uint32_t base_bits = bit_width_u32((job.in_size + 31u) / 32u);
            const uint32_t dsize_bits = bit_width_u32(bit_width_u32(job.in_size - 1u));
            const uint32_t base_size = (uint32_t)(h >> 8) & ((1u << base_bits) - 1u);
            const uint32_t delta_bits = (uint32_t)(h >> (8u + base_bits)) & ((1u << dsize_bits) - 1u);
            const uint32_t table_at = 8u + base_bits + dsize_bits;
            const uint32_t bit = table_at + sl * delta_bits;
            const uint32_t wi = (bit >> 5) * 4u;
            const uint64_t d = (uint64_t)br_load(job, wi) | ((uint64_t)br_load(job, wi + 4u) << 32);
            const uint32_t delta = (uint32_t)(d >> (bit & 31u)) & ((1u << delta_bits) - 1u);
            my_len = base_size + delta;
            hdr_bytes = ((table_at + 32u * delta_bits + 31u) / 32u) * 4u;
        }
        const uint32_t incl = wave::half_scan_incl(my_len);
        if (start) br.init(job.in, job.in_limit, hdr_bytes + incl - my_len);
    }
    clk.lap(kPhSetup);
    // one copy of the table builder in the instruction stream, run three times (ICP, distance, literal):
    // inlined three times it was most of the kernel's code size, beyond what the instruction cache holds
    tables_ok = true;
#pragma nounroll
    for (uint32_t k = 0; k < 3u; ++k) {
        const bool ok = build_table(table_of(L, k, far_syms), build_lens(L, k), br, start, sl);
        tables_ok = tables_ok && ok;
    }
    return start;
}

// ---- stage: the commands of a round (PageDecoder.cpp:290-320, :338-404; format A.6 step 1, A.7, A.8).
struct RoundCommands {
    uint32_t sent_mask;     // lanes of the half that decoded the sentinel (704): the page's last round
    uint32_t n;             // real commands of the round (0..32)
    bool     is_cmd;        // this lane holds one
    uint32_t ins, copy;     // insert and copy length (copy 0: insert-only command)
    uint32_t dcode;         // distance code (0 = implicit "last distance")
    uint32_t dist;          // distance for explicit codes >= 16; ring codes are resolved by resolve_distance_ring
};
// One command per lane.  Two refill points per command: with >= 32 bits in the window the command symbol (<= 15 bits)
// leaves >= 17 for the insert/copy extra bits, and likewise the distance symbol for its extra bits; longer fields
// (rare) take the general read.
template <class Lds, class Reader, class Clock>
__device__ __forceinline__ RoundCommands decode_round_commands(const Lds& L, const uint32_t* len_code_tab, const TableRef& t_icp, const TableRef& t_dist,
                                                                Reader& br, bool live, uint32_t sl, Clock& clk)
{
    RoundCommands c;
    uint32_t sym = 0, len = 0;
    if (live) { br.ensure(32); sym = decode_symbol<kLutBitsIcp>(t_icp, br, len); }
    clk.lap(kPhCmdSym);
    c.sent_mask = wave::half_of(wave::ballot_eq_k<kSentinel>(sym));          // (sym stays 0 in a half without a page)
    c.n = c.sent_mask ? ctz_u32(c.sent_mask) : 32u;
    c.is_cmd = live && sl < c.n;
    if (live && sl <= c.n) br.consume(len);                           // the sentinel's own bits are consumed too
    c.ins = 0; c.copy = 0; c.dist = 0; c.dcode = 0;
    if (c.is_cmd) {
        // insert and copy length codes (for insert-only symbols 705..727 the copy length stays 0)
        const bool has_copy = sym < kSentinel;
        const uint32_t cell = sym >> 6;
        const uint32_t ic = has_copy ? ((0x298500u >> (2u * cell)) & 3u) * 8u + ((sym >> 3) & 7u) : min_u32(sym - kSentinel, 23u);
        const uint32_t cc = ((0x262444u >> (2u * cell)) & 3u) * 8u + (sym & 7u);
        const uint32_t it = len_code_tab[ic], ct = has_copy ? len_code_tab[24u + cc] : 0u;
        const uint32_t ie = it >> 16, ce = ct >> 16;
        uint32_t xi, xc;
        if (ie + ce <= 17u) {                                       // both fields are already in the window
            const uint32_t x = br.peek(ie + ce);
            br.consume(ie + ce);
            xi = x & ((1u << ie) - 1u); xc = x >> ie;
        } else { xi = br.read(ie); xc = br.read(ce); }
        c.ins = (it & 0xFFFFu) + xi;
        c.copy = has_copy ? (ct & 0xFFFFu) + xc : 0u;
        clk.lap(kPhCmdExtra);
        if (has_copy && sym >= 128u) {                              // explicit distance symbol
            uint32_t dl;
            br.ensure(32);
            c.dcode = decode_symbol<kLutBitsDist>(t_dist, br, dl);
            br.consume(dl);
            if (c.dcode >= 16u) {                                   // PageDecoder.cpp:365-390
                const uint32_t pp = L.page_params;
                const uint32_t npostfix = pp & 3u, ndirect = (pp >> 8) & 0xFFu;
                if (c.dcode < 16u + ndirect) c.dist = c.dcode - 15u;
                else {
                    const uint32_t x = c.dcode - ndirect - 16u;
                    const uint32_t nbits = min_u32(1u + (x >> (npostfix + 1u)), 24u);
                    uint32_t extra;
                    if (nbits <= 17u) { extra = br.peek(nbits); br.consume(nbits); } else extra = br.read(nbits);
                    const uint32_t hcode = x >> npostfix, lcode = x & ((1u << npostfix) - 1u);
                    c.dist = ((((2u + (hcode & 1u)) << nbits) - 4u + extra) << npostfix) + lcode + ndirect + 1u;
                }
            }
        }
    }
    return c;
}

// ---- stage: the distance ring (PageDecoder.cpp:345-364, :396-403): the last four distances pushed, most recent first.
// It lives in LDS as a circular buffer of eight words: the t-th distance pushed in the page (t counts from 4: the four initial
// entries 16, 15, 11, 4 are pushes 0..3) sits in word t & 7, and all a lane keeps is the page's push count so far.  The q-th most
// recent push before a round is word (T - 1 - q) & 7 -- ONE LDS read per lane, for the lanes that need a carried entry at all --
// and a round stores its last four pushes in words T .. T + cnt - 1 (& 7): they cannot meet the four words below T that the same
// round still reads (eight consecutive push numbers at most).  Rounds 1-3 kept the four entries in registers and folded the previous
// round's pushes in with a chain of selects on the push count (sixteen v_cndmask a round, on a kernel bound by the vector ALU).
struct DistanceRing {
    uint32_t total = 4;                             // pushes of the page so far, the four initial entries included
    template <class Lds> __device__ __forceinline__ void reset(Lds& L, bool starting, uint32_t sl)
    {
        // 4, 11, 15, 16 most recent first (PageDecoder.cpp:150-153) = pushes 3, 2, 1, 0; one word per lane out of a packed constant
        // (four constants become a constant vector that is kept in registers for the whole kernel, spilled, and reloaded every round)
        if (starting && sl < 4u) L.ring[sl] = (0x040B0F10u >> (8u * sl)) & 0xFFu;
        if (starting) total = 4u;
    }
};
struct RingWords {};                                // (rounds 1-3: the ring words, loaded at the top of a round)
template <class Lds>
__device__ __forceinline__ RingWords load_ring_pushes(const Lds&, const DistanceRing&) { return RingWords{}; }
// Codes 1..15 are resolved in command order; explicit distances and code 0 need no serial step.  On return c.dist
// is final for every copy command of the round.
template <class Lds>
__device__ __forceinline__ void resolve_distance_ring(Lds& L, DistanceRing& ring, const RingWords&, RoundCommands& c, uint32_t sl)
{
    const uint32_t T = ring.total;
    const uint32_t dcode = c.dcode;
    uint32_t dist = c.dist;
    const bool is_copy = c.is_cmd && c.copy > 0u;
    const uint32_t push_mask = wave::half_of(wave::ballot_ne0(dcode));        // (a distance code is only decoded for a command with a copy)
    // A code 1..15 refers to the r-th most recent push before the command (r from the code): either
    // a command of this round (lane `src`) or the ring carried in from earlier rounds.  All lanes
    // whose source is already known resolve together; a chain of ring codes takes one pass per link
    // (the lowest unresolved lane is always resolvable).  Code 0 ("the last distance") is r = 0 without a push: it waits
    // until the chains are done.
    uint32_t pend = wave::half_of(wave::ballot_lt_k<15u>(dcode - 1u));        // codes 1 .. 15
    const uint32_t r = dcode < 4u ? dcode : (dcode < 10u ? 0u : 1u);
    const uint32_t below0 = push_mask & ((1u << sl) - 1u);
    uint32_t below = below0;
    const uint32_t cnt = (uint32_t)__popc(below);
    // the carried entry r - cnt (when the round has fewer than r + 1 pushes before me): requested now, used in the loop
    const uint32_t carried = L.ring[(T - 1u - (r - cnt)) & 7u];
    {
        if (r >= 1u && below) below &= ~(1u << msb_u32(below));
        if (r >= 2u && below) below &= ~(1u << msb_u32(below));
        if (r >= 3u && below) below &= ~(1u << msb_u32(below));
        const bool from_round = r < cnt;
        const uint32_t src = from_round ? msb_u32(below) : 0u;
        const uint32_t j = dcode >= 4u ? (dcode - 4u) % 6u : 0u, mag = dcode >= 4u ? (j >> 1) + 1u : 0u;
        while (wave::any(pend != 0u)) {
            const bool mine = ((pend >> sl) & 1u) != 0u;
            const bool ready = mine && (!from_round || ((pend >> src) & 1u) == 0u);
            const uint32_t from = wave::half_shfl(dist, src);
            if (ready) {
                const uint32_t val = from_round ? from : carried;
                dist = (j & 1u) ? val + mag : val - mag;
            }
            pend &= ~wave::half_ballot(ready);
        }
    }
    {
        const uint32_t from = wave::half_shfl(dist, below0 ? msb_u32(below0) : 0u);
        if (is_copy && dcode == 0u) dist = below0 ? from : carried;     // (r = 0, cnt = 0: `carried` is the most recent push of earlier rounds)
        // the round's last four pushes go to the ring
        const bool pusher = is_copy && dcode != 0u;
        const uint32_t above = (uint32_t)__popc((push_mask >> sl) >> 1);    // pushes after mine
        const uint32_t pushes = (uint32_t)__popc(push_mask);
        if (pusher && above < 4u) L.ring[(T + pushes - 1u - above) & 7u] = dist;
        ring.total = T + pushes;
    }
    c.dist = dist;
}

// The persistent page loop of one wavefront.  Each 32-lane half decodes its own page and takes the
// next page from the work counter as soon as it is done, independently of the other half: pages
// differ a lot in their number of rounds (stored, run-length and text pages side by side), and a
// half that waited for its neighbour would idle for the difference.  The wavefront's control flow
// stays uniform: one iteration = (page start for the halves that need one) + (one round for the
// halves inside a page) + (page end for the halves whose page just finished), each under per-half
// predicates.
// which LDS record a lane works in, and with which geometry: a record per half, or (one page per wavefront) one record for the whole
// wavefront in the storage of both -- the upper half has no page of its own and never writes to it except as a member of a copy team
template <bool kSolo> struct PageRecord;
template <> struct PageRecord<false> {
    typedef GeoPair G;
    static __device__ __forceinline__ PageLds& of(WaveLds& W, uint32_t lane) { return W.page[lane >> 5]; }
};
template <> struct PageRecord<true> {
    typedef GeoSolo G;
    static __device__ __forceinline__ PageLdsSolo& of(WaveLds& W, uint32_t) { return *reinterpret_cast<PageLdsSolo*>(&W.page[0]); }
};

template <bool kProf, bool kSolo>
__device__ inline void decode_pages(WaveLds& W, const DecodeArgs& a, unsigned long long* prof_lds)
{
    typedef typename PageRecord<kSolo>::G G;
    PhaseClock<kProf> clk;
    clk.start(prof_lds);
    const uint32_t lane = wave::lane_id();
    const uint32_t sl = lane & 31u;
    PageLdsT<G>& L = PageRecord<kSolo>::of(W, lane);

    // the three prefix codes of a page: ICP, distance, literal (PageDecoder.cpp:125-147)
    uint16_t* const far_syms = a.far_syms + (size_t)blockIdx.x * (2u * kFarSymStride);
    const TableRef t_icp{L.lut_icp, L.sorted_icp, L.limit[0], L.first_offs[0], kIcpAlphabet, kLutBitsIcp, far_syms};
    const TableRef t_dist{L.lut_dist, L.sorted_dist, L.limit[1], L.first_offs[1], kDistAlphabet, kLutBitsDist, far_syms};
    const TableRef t_lit{L.lut_lit, L.sorted_lit, L.limit[2], L.first_offs[2], kLitAlphabet, kLutBitsLit, nullptr};

    const uint32_t resync_quarters = a.status[3];                       // pairing policy, set by the schedule kernel
    // ---- per-half state of the page under construction
    PageJob job = no_job(a);
    bool live = false;               // inside a compressed page
    // kSolo (chosen per wavefront by decode_kernel_body): this wavefront decodes one page at a time, its upper half takes no pages
    // and helps with long copies instead.  A template parameter, not a flag: the two-page instantiation is compiled without it.
    constexpr bool solo = kSolo;
    bool finished = lane >= 32u && solo;    // the work counter ran out for this half (or it sits this launch out)
    BitReaderT<(BROTLIG_EXP_GLDS == 1)> br;
    br.base = a.in; br.limit8 = 0; br.buf = 0; br.avail = 64; br.next = 0; br.queue = 0; br.queued = 64; br.flight = 0; br.zero = wave::opaque_zero();
    br.slot = nullptr; br.slot0 = 0;
#if BROTLIG_EXP_GLDS
    br.slot = W.glds + 16u * lane;
    br.slot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)W.glds);      // (LDS addresses are the low 32 bits of the generic ones)
#endif
    DistanceRing ring;
    uint32_t out_pos = 0;            // bytes of the page produced so far
    uint32_t prev_tail = 0;          // literals decoded but not yet consumed
    uint32_t carry_head = 0;
    bool bad = false;
    OutView view{L.win, 0u};
    uint32_t flushed = 0;            // page bytes below this are in global memory

    for (;;) {
        // ---- page start.  A half without a page takes one -- unless the other half is within
        //      resync_quarters / 4 of finishing its own page: then it waits and both start together (one
        //      joint table build instead of two single ones).  The schedule kernel sets the threshold per
        //      launch: 1 when neighbouring pages differ in cost (a free half starts over at once), 4 when
        //      they are alike -- then the halves stay in step, which keeps rounds of the same shape
        //      paired (measured on the BC3 config: 7 % faster in step than out of phase).
        {
            const uint32_t near_end = (live && (job.out_size - out_pos) * 4u < job.out_size * resync_quarters) ? 1u : 0u;
            const uint32_t other_near = wave::other_half(near_end);
            const bool want = !live && !finished && other_near == 0u;
            if (wave::any(want)) {
                clk.lap(kPhDelta);
                bool tables_ok = true;
                const bool start = start_pages(a, L, job, br, want, finished, sl, far_syms, tables_ok, [](const PageJob&) {}, clk);
                ring.reset(L, start, sl);
                if (start) {
                    out_pos = 0; prev_tail = 0; carry_head = 0; flushed = 0; bad = false;
                    view.win_base = 0u;
                    live = true;
                    // an undefined code description rejects the page: with the page "full" its first round is refused
                    // (or is a bare sentinel), nothing is assembled or flushed, and the page ends with `bad` set
                    if (!tables_ok) { bad = true; out_pos = flushed = job.out_size; }
                    if (kAblate & kAblRounds) live = false;             // (timing build: what the page starts alone cost)
                }
                clk.lap(kPhTables);
            }
        }
        if ((kAblate & kAblRounds) && wave::any(!finished)) continue;
        if (!wave::any(live)) break;                                    // a half without a page has none left to take
        const bool in_page = live;

        // ---- rounds (PageDecoder.cpp:174-236; format A.6) until a page ends
        do {
        // -- 1. one command per lane (the previous round's ring pushes are requested first: they are needed in step 2)
        const RingWords pushed = load_ring_pushes(L, ring);
        RoundCommands cmd = decode_round_commands(L, W.len_code_tab, t_icp, t_dist, br, live, sl, clk);
        const uint32_t sent_mask = cmd.sent_mask, n = cmd.n;
        const bool is_cmd = cmd.is_cmd;

        clk.lap(kPhCommands);
        clk.count(kPhRounds, 1);
        if constexpr (kProf) {                                          // rounds in which one half has no page left
            const uint64_t lm = wave::ballot64(live);
            clk.count(kPhSlow, (((uint32_t)lm != 0u) != ((uint32_t)(lm >> 32) != 0u)) ? 1u : 0u);
        }
        // -- 2. distance ring
        resolve_distance_ring(L, ring, pushed, cmd, sl);
        const uint32_t ins = cmd.ins, copy = cmd.copy, dist = cmd.dist;

        clk.lap(kPhRing);
        // -- 3. output positions
        const uint32_t tot = ins + copy;
        const uint32_t incl_tot = wave::half_scan_incl(tot);
        const uint32_t incl_ins = wave::half_scan_incl(ins);
        const uint32_t round_bytes = wave::half_bcast(incl_tot, 31);
        const uint32_t litcount = wave::half_bcast(incl_ins, 31);
        const uint32_t cmd_out = out_pos + incl_tot - tot;              // first literal of my command
        const uint32_t copy_dst = cmd_out + ins;
        // (every command emits at least one byte, so a page of full rounds ends here after out_size / 32 rounds at most)
        if (live && round_bytes > job.out_size - out_pos) { bad = true; live = false; }
        const bool ok_cmd = is_cmd && live;

        // literal bookkeeping of the round (PageDecoder.cpp:196-199)
        const uint32_t lit_a = incl_ins - ins;                          // my literals are consumption indices [lit_a, lit_a + ins)
        const uint32_t rel0 = incl_tot - tot;                           // my first byte, relative to the round
        const uint32_t ac = litcount > prev_tail ? litcount - prev_tail : 0u;
        const uint32_t mult = (live && n) ? div_small(min_u32(ac, 0x200000u) + n - 1u, n) : 0u;
        const uint32_t rlit = n * mult;                                 // literals decoded this round (0 when !live)
        uint32_t next_j = sl;                                           // next literal of the round this lane decodes
        const bool dist_ok = dist != 0u && dist <= copy_dst;
        if (ok_cmd && copy > 0u && !dist_ok) bad = true;
        const bool cp = ok_cmd && copy > 0u && dist_ok;
        clk.lap(kPhPositions);

        // The round's output is assembled in the LDS window in byte ranges ("groups") of at most
        // kRoundMax bytes -- nearly always a single group.  A command that crosses a group boundary
        // contributes a piece to each group; a copy piece past the first is an ordinary copy from
        // `dist` bytes back (its earlier bytes are final by then).
        const uint32_t ngroups = live ? (round_bytes + G::kRoundMax - 1u) / G::kRoundMax : 0u;
        const bool multi_group = wave::any(ngroups > 1u);
        const uint64_t okcmd_w = wave::ballot64(ok_cmd), cp_w = wave::ballot64(cp);     // (lane masks: see wave::ballot_gt)
        for (uint32_t g = 0; ; ++g) {
            const uint64_t on_w = wave::ballot_lt(g, ngroups);
            if (on_w == 0ull) break;
            const bool on = wave::from_mask(on_w);
            const uint32_t g0 = g * G::kRoundMax, g1 = on ? min_u32(round_bytes, g0 + G::kRoundMax) : g0;
            const uint32_t gpos = out_pos + g0;                         // page position of the group's first byte

            // -- 3b. flush the finished bytes, slide the window when the group does not fit
            flush_and_slide<G>(view, flushed, job.out, on, on_w, gpos, out_pos + g1, sl);
            clk.lap(kPhSlide);
            clk.count(kPhGroups, 1);
            clk.halves(kPhGroupHalves, on);
            const uint32_t span0 = gpos - view.win_base;                // window index of the group's first byte

            // -- 3c. my pieces in this group
            const uint32_t cs = rel0 + ins;                             // my copy starts here (round-relative)
            uint64_t in_group_w;                                        // lanes with a piece in the group
            uint32_t la, nlit, lit_f, plen, pdst;                       // my first byte in the group; my literal bytes in it and the consumption
                                                                        // index of the first; my copy bytes in it and their page position
            if (multi_group) {
                in_group_w = on_w & okcmd_w & wave::ballot_lt(rel0, g1) & wave::ballot_gt(rel0 + tot, g0);
                const bool in_group = wave::from_mask(in_group_w);
                const uint32_t lb = cs < g1 ? cs : g1;
                la = rel0 > g0 ? rel0 : g0;
                nlit = (in_group && lb > la) ? lb - la : 0u;
                lit_f = lit_a + (la - rel0);
                const uint32_t ca = cs > g0 ? cs : g0, cb = rel0 + tot < g1 ? rel0 + tot : g1;
                plen = (in_group && cp && cb > ca) ? cb - ca : 0u;
                pdst = out_pos + ca;
            } else {                                                    // the round is one group (nearly always): every command lies in it whole
                in_group_w = on_w & okcmd_w;
                la = rel0;
                nlit = wave::from_mask(in_group_w) ? ins : 0u;
                lit_f = lit_a;
                plen = wave::from_mask(on_w & cp_w) ? copy : 0u;
                pdst = copy_dst;
            }
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
            // Round 4: the same for a short piece whose source lies in the window but wholly below the group (final before the
            // group started -- 27 % of the copy pieces of the mixed data, most of the first dependency level): read ahead from LDS
            // into the same registers and stored with the far pieces, instead of a level of its own.
            const bool near_direct = BROTLIG_TUNE_EARLY_NEAR && plen != 0u && far_len == 0u && plen <= kShortCopy && dist >= plen &&
                                     src_end <= gpos && !(kAblate & kAblLevels);
            const FarSources far = fetch_far_sources(job.out, L.win, psrc - view.win_base, near_direct, plen, psrc, far_len, sl);
            const bool far_direct = far.direct;
            const uint32_t stage_off = far.stage_off;
            clk.lap(kPhPieces);
            // literals of the group: consumption indices [F0, F1)
            const uint32_t mine_before = (on && ok_cmd) ? (cs <= g0 ? ins : (rel0 < g0 ? g0 - rel0 : 0u)) : 0u;   // my literals before g0
            uint32_t F0 = 0, F1 = litcount;                             // single group: all of the round's literals
            if (multi_group) { F0 = wave::half_sum(mine_before); F1 = F0 + wave::half_sum(nlit); }
            // exact dependencies of my copy piece: the pieces (of commands before me) that own bytes of its source range
            // inside this group; everything below the group is final
            const uint32_t dep_mask = piece_dependencies<PhaseClock<kProf>, G>(L.start_bits, L.start_cum, on, in_group_w, la - g0, gpos,
                                                         psrc, src_end, plen != 0u && !far_direct && !(kAblate & kAblDeps), sl, clk);
            clk.lap(kPhCopyFence);

            // -- 4. literals of the group.  Literal j of the round comes from sub-stream j mod 32 and is
            //       consumption index prev_tail + j (PageDecoder.cpp:196-206); indices below prev_tail were
            //       decoded in earlier rounds and wait in the carry ring.  They are laid down in consumption
            //       order (the reference's literal queue, PageDecoder.cpp:164-166,:209-211, one group at a time), in
            //       the staging area, which is free until the far sources are stored; then every command moves its
            //       own run to the window like a short copy.
            uint8_t* const lits = reinterpret_cast<uint8_t*>(L.stage);
            if (on) {
                const uint32_t cf1 = F1 < prev_tail ? F1 : prev_tail;                   // carried part of [F0, F1)
                for (uint32_t f = F0 + sl; f < cf1; f += 32u) lits[f - F0] = L.carry[(carry_head + f) & 63u];
                // the last group also decodes the literals beyond what the round consumes (fewer than 32):
                // they wait in the carry ring for the next round
                const bool last_group = g + 1u == ngroups;
                const uint32_t J1 = last_group ? rlit : (F1 > prev_tail ? F1 - prev_tail : 0u);
                const uint32_t keep_at = carry_head + prev_tail;        // ring index of consumption index `litcount` (mod 64)
                auto place = [&](uint32_t j, uint32_t lit) {
                    const uint32_t f = prev_tail + j;
                    if (f < litcount) lits[f - F0] = (uint8_t)lit;
                    else L.carry[(keep_at + (f - litcount)) & 63u] = (uint8_t)lit;
                };
                // two literals per refill check while at least two are left (a literal is at most 15 bits)
                for (; next_j + 32u < J1; next_j += 64u) {
                    uint32_t l0, l1;
                    clk.count(kPhLitSteps, 1);
                    br.ensure(30);
                    const uint32_t lit0 = decode_symbol<kLutBitsLit>(t_lit, br, l0);
                    br.consume(l0);
                    const uint32_t lit1 = decode_symbol<kLutBitsLit>(t_lit, br, l1);
                    br.consume(l1);
                    place(next_j, lit0);
                    place(next_j + 32u, lit1);
                }
                if (next_j < J1) {
                    uint32_t ll;
                    br.ensure(15);
                    const uint32_t lit = decode_symbol<kLutBitsLit>(t_lit, br, ll);
                    br.consume(ll);
                    place(next_j, lit);
                    next_j += 32u;
                }
            }
            wave::sync();
            // -- 4b. literal runs: from the queue to their place in the window (own lane; long inserts in teams)
            if (!(kAblate & kAblLitStore)) {
                const uint32_t q_idx = lit_f - F0, w_idx = span0 - g0 + la;
                own_copy_simple(lits + q_idx, L.win + w_idx, nlit, wave::ballot_lt_k<kOwnCopy>(nlit - 1u));      // 1 <= nlit <= kOwnCopy
                const uint64_t long_w = wave::ballot_gt_k<kOwnCopy>(nlit);
                if (long_w != 0ull) {
                    const uint32_t lmask = wave::half_of(long_w);
                    const Team tl = make_team(lmask, sl);
                    const uint32_t l_src = wave::half_shfl(q_idx, tl.job), l_dst = wave::half_shfl(w_idx, tl.job);
                    const uint32_t l_len = wave::half_shfl(nlit, tl.job);
                    const bool act = tl.serves && lmask != 0u;
                    for (uint32_t c = tl.member; wave::any(act && 8u * c < l_len); c += 1u << tl.log2_size) {
                        const uint32_t j = 8u * c;
                        if (act && j < l_len) store_bytes(L.win + l_dst + j, load_u64u(lits + l_src + j), l_len - j);
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

            // -- 5b. LZ77 copies in dependency levels.  The levels are the longest dependent chain of a round (LDS read -> LDS
            //        write -> ballot, three to four times over): while a wave is in them it goes first at its SIMD's issue
            //        port (s_setprio; +1.2 .. 1.5 % measured, any level 1..3; raised around the command decode as well it is
            //        the same on mixed data and +0.6 % on text, around the whole group loop it loses)
            wave::set_priority(1);
#if BROTLIG_TUNE_PLAIN_LEVELS
            {   // one question per group instead of three per level: does any piece need more than the plain own-lane batch?
                const uint64_t plain_w = (wave::ballot_eq0(far_len) | wave::ballot_eq(far_len, pattern)) & ~wave::ballot_lt(dist, plen) & wave::ballot_lt_k<33u>(plen);      // simple, and one batch
                if (!kAblate && (wave::ballot_ne0(plen) & ~far.direct_w & ~plain_w) == 0ull)
                    copy_levels_plain(L.win, L.stage, plen, far_len, stage_off, src_idx, dst_idx, far.direct_w, dep_mask, sl, clk);
                else
                    copy_levels(L.win, L.stage, plen, dist, far_len, stage_off, src_idx, dst_idx, far.direct_w, dep_mask, sl, solo, clk);
            }
#else
            copy_levels(L.win, L.stage, plen, dist, far_len, stage_off, src_idx, dst_idx, far.direct_w, dep_mask, sl, solo, clk);
#endif
            wave::set_priority(0);
            clk.lap(kPhCopyLevels);
        }

        // carry ring bookkeeping: consumed entries leave at the head; when the round consumed fewer
        // literals than were waiting (rlit == 0 then), the rest stays where it is
        if (live) {
            const uint32_t new_head = carry_head + min_u32(prev_tail, litcount);
            carry_head = new_head;
            prev_tail = rlit + prev_tail - litcount;
        }

        if (live) out_pos += round_bytes;                               // (a rejected round produced nothing)
        if (sent_mask) live = false;
        } while (!wave::any(in_page && !live));

        // ---- page end for the halves whose page finished (or was rejected) in the last round
        const bool ended = in_page && !live;
        wave::sync();
        if (ended) flushed = flush_window(job.out, view, flushed, out_pos, true, sl);
        if (ended && out_pos != job.out_size) bad = true;                // a valid page fills its output exactly

    // ---- per-page delta decode of the colour sub-streams
    delta_decode_page(job, ended && (L.page_params >> 16) != 0u && !bad, sl);
    if (ended && bad && sl == 0u) flag_bad_page(a, L.page_stream);
    }
    clk.lap(kPhDelta);
    clk.flush(a.prof, lane);
}

}  // namespace brotlig
