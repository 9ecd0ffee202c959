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
    uint32_t ring_push[2][4] __attribute__((aligned(16)));
};
static_assert(kLutBitsIcp == 8 && kLutBitsDist == 8 && kLutBitsLit == 8, "build areas are laid out for three 512-byte LUTs");
static_assert(__builtin_offsetof(EntropyLds, build_tail) == 1536 && __builtin_offsetof(EntropyLds, lit_lens) == 1536 + 544, "build areas must follow the LUTs");
static_assert(512 + 544 >= kIcpAlphabet && 544 >= kDistAlphabet, "code lengths fit their build areas");

__device__ __forceinline__ uint8_t* build_lens(EntropyLds& L, uint32_t k)
{
    return k == 0u ? reinterpret_cast<uint8_t*>(L.lut_lit) : k == 1u ? L.build_tail : L.lit_lens;
}

struct __attribute__((aligned(16))) EntropyWaveLds {
    EntropyLds page[2];
    uint32_t len_code_tab[48];
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
    BitReader br;
    br.base = a.in; br.limit8 = 0; br.buf = 0; br.avail = 64; br.next = 0; br.queue = 0; br.queued = 64; br.flight = 0; br.zero = wave::opaque_zero();
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
                if (start) {
                    ring.reset();
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
        const Bytes16 pushed = load_ring_pushes(L, ring);
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

            // -- 3b. flush, and make room in the window when the group does not fit.  Every group first stores the
            //        finished bytes below it (aligned 16-byte pieces; `flushed` is 16-byte aligned until the page's
            //        last flush and at most kRoundMax + 15 bytes behind), so that a far copy -- source below the
            //        window, i.e. more than kHist >= kRoundMax + 16 bytes back -- only ever reads global memory
            //        written by an EARLIER group's flush.  The slide keeps kHist .. kHist + 15 bytes of history and
            //        brings them down in one step, all reads before the writes.
            const bool slide = on && out_pos + g1 > view.win_base + kWin && !(kAblate & kAblSlide);
            wave::sync();
            {
                const uint32_t e16 = gpos & ~15u;
                const uint32_t p0 = flushed + 16u * sl, p1 = p0 + 512u;
                const bool f0 = on && p0 < e16, f1 = on && p1 < e16;
                Bytes16 a0 = {0u, 0u, 0u, 0u}, a1 = a0;
                if (f0) a0 = load16(view.win + (p0 - view.win_base));
                if (f1) a1 = load16(view.win + (p1 - view.win_base));
                if (f0) store16(job.out + p0, a0);
                if (f1) store16(job.out + p1, a1);
                if (on && e16 > flushed) flushed = e16;
            }
            if (wave::any(slide)) {
                const uint32_t nb = slide ? (gpos - kHist) & ~15u : view.win_base;
                const uint32_t shift = nb - view.win_base, count = shift ? gpos - nb : 0u;
                const uint32_t i0 = 16u * sl, i1 = 512u + 16u * sl;
                Bytes16 m0 = {0u, 0u, 0u, 0u}, m1 = m0;
                if (i0 < count) m0 = load16(view.win + shift + i0);
                if (i1 < count) m1 = load16(view.win + shift + i1);
                wave::sync();
                if (i0 < count) store16(view.win + i0, m0);
                if (i1 < count) store16(view.win + i1, m1);
                view.win_base = nb;
            }
            wave::sync();
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
            const bool far_direct = far_len != 0u && far_len == plen && plen <= kShortCopy;
            const bool staged = far_len != 0u && !far_direct;
            const uint32_t stage_len = staged ? (far_len + 7u) & ~7u : 0u;
            const bool any_staged = wave::any(staged);
            uint32_t stage_off = 0;                                     // 8-byte aligned offset into L.stage
            if (any_staged) stage_off = wave::half_scan_incl(stage_len) - stage_len;
            const bool far_teams = any_staged && wave::any(staged && far_len > kShortCopy);
            uint64_t fe0 = 0, fe1 = 0, fe2 = 0, fe3 = 0, te0 = 0, te1 = 0;
            Team ft{5u, 0u, 0u, false};
            uint32_t ft_src = 0, ft_len = 0, ft_stage = 0;
            const uint32_t clip8 = plen >= 8u ? plen - 8u : 0u;
            if (far_len != 0u && (far_direct || !far_teams)) {
                const uint8_t* s8 = job.out + psrc;
                const uint32_t lim = far_direct ? clip8 : 24u;          // a staged piece keeps plain offsets
                fe0 = load_u64u(s8);
                if (far_len > 8u) fe1 = load_u64u(s8 + min_u32(8u, lim));
                if (far_len > 16u) fe2 = load_u64u(s8 + min_u32(16u, lim));
                if (far_len > 24u) fe3 = load_u64u(s8 + min_u32(24u, lim));
            }
            if (far_teams) {
                const uint32_t staged_mask = wave::half_ballot(staged);
                ft = make_team(staged_mask, sl);
                ft_src = wave::half_shfl(psrc, ft.job); ft_len = wave::half_shfl(far_len, ft.job);
                ft_stage = wave::half_shfl(stage_off, ft.job);
                ft.serves = ft.serves && staged_mask != 0u;
                if (ft.serves && 8u * ft.member < ft_len) te0 = load_u64u(job.out + ft_src + 8u * ft.member);
                if (ft.serves && 8u * (ft.member + (1u << ft.log2_size)) < ft_len) te1 = load_u64u(job.out + ft_src + 8u * (ft.member + (1u << ft.log2_size)));
            }
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
            const uint32_t piece_mask = wave::half_ballot(in_group);
            if (on && sl < kRoundMax / 32u) L.start_bits[sl] = 0u;
            wave::sync();
            if (in_group) {
                const uint32_t b = (rel0 > g0 ? rel0 : g0) - g0;        // my first byte in the group
                atomicOr(&L.start_bits[b >> 5], 1u << (b & 31u));
            }
            wave::sync();
            {
                const bool rd = on && sl < kRoundMax / 32u;
                const uint32_t w = rd ? L.start_bits[sl] : 0u;
                const uint32_t cw = wave::half_scan_incl((uint32_t)__popc(w));
                if (rd) L.start_cum[sl] = (uint8_t)(cw - (uint32_t)__popc(w));
            }
            wave::sync();
            clk.lap(kPhBitmaps);
            // exact dependencies of a source range [s0, s1) of mine: the pieces (of commands before me) that own bytes
            // of it inside this group; everything below the group is final
            const uint32_t first_piece = ctz_u32(piece_mask);
            auto deps_of = [&](uint32_t s0, uint32_t s1) -> uint32_t {
                uint32_t m = 0;
                if (s1 > gpos) {
                    const uint32_t hi_rel = s1 - 1u - gpos;
                    const uint32_t hi = L.start_cum[hi_rel >> 5] + (uint32_t)__popc(L.start_bits[hi_rel >> 5] & (0xFFFFFFFFu >> (31u - (hi_rel & 31u))));
                    uint32_t lo = 0;
                    if (s0 > gpos) {
                        const uint32_t lo_rel = s0 - gpos;
                        lo = L.start_cum[lo_rel >> 5] + (uint32_t)__popc(L.start_bits[lo_rel >> 5] & (0xFFFFFFFFu >> (31u - (lo_rel & 31u)))) - 1u;
                    }
                    // ranks lo .. hi-1 among the group's pieces; the pieces are consecutive commands (every
                    // command has at least one byte), so rank r is lane first_piece + r.  Only pieces before
                    // me can still be unfinished.
                    const uint32_t lo_l = first_piece + lo, hi_l = min_u32(first_piece + hi, sl);
                    if (hi_l > lo_l) m = ((1u << hi_l) - 1u) & ~((1u << lo_l) - 1u);
                }
                return m;
            };
            uint32_t dep_mask = (plen && !(kAblate & kAblDeps)) ? deps_of(psrc, src_end) : 0u;
            // Forwarding: a piece that does not overlap itself and whose whole source lies inside ONE earlier piece of
            // the same kind (a plain copy inside the window) reads that piece's source instead of its output -- the
            // same bytes, one dependency level earlier (chains of copies of copies are a fifth of all levels on mixed
            // data, two fifths on records).  Its dependencies are then those of the new range.
            uint32_t fsrc = psrc;
            if (kForwardHops != 0u && !(kAblate & kAblDeps)) {
                const bool plain = plen != 0u && dist >= plen && far_len == 0u;
                const uint32_t plain_mask = wave::half_ballot(plain);
#pragma nounroll
                for (uint32_t hop = 0; hop < kForwardHops; ++hop) {
                    const bool single = plain && dep_mask != 0u && (dep_mask & (dep_mask - 1u)) == 0u && ((plain_mask & dep_mask) != 0u);
                    if (!wave::any(single)) break;
                    const uint32_t d = single ? ctz_u32(dep_mask) : 0u;
                    const uint32_t d_dst = wave::half_shfl(pdst, d), d_len = wave::half_shfl(plen, d), d_src = wave::half_shfl(fsrc, d);
                    if (single && fsrc >= d_dst && fsrc + plen <= d_dst + d_len) {
                        fsrc = d_src + (fsrc - d_dst);
                        dep_mask = deps_of(fsrc, fsrc + plen);
                    }
                }
            }
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
            const uint32_t src_idx = fsrc - view.win_base;              // window index of the pattern start (negative when far)
            const uint32_t dst_idx = pdst - view.win_base;
            if (far_direct) {
                uint8_t* d = L.win + dst_idx;
                if (plen >= 8u) {
                    __builtin_memcpy(d, &fe0, 8);
                    if (plen > 8u) __builtin_memcpy(d + min_u32(8u, clip8), &fe1, 8);
                    if (plen > 16u) __builtin_memcpy(d + min_u32(16u, clip8), &fe2, 8);
                    if (plen > 24u) __builtin_memcpy(d + clip8, &fe3, 8);
                } else store_bytes(d, fe0, plen);
            }
            if (any_staged) {
                if (!far_teams) {
                    if (staged) {
                        uint64_t* st = &L.stage[stage_off >> 3];
                        st[0] = fe0;
                        if (far_len > 8u) st[1] = fe1;
                        if (far_len > 16u) st[2] = fe2;
                        if (far_len > 24u) st[3] = fe3;
                    }
                } else {
                    const uint32_t tsz = 1u << ft.log2_size;
                    if (ft.serves && 8u * ft.member < ft_len) L.stage[(ft_stage >> 3) + ft.member] = te0;
                    if (ft.serves && 8u * (ft.member + tsz) < ft_len) L.stage[(ft_stage >> 3) + ft.member + tsz] = te1;
                    for (uint32_t c = ft.member + 2u * tsz; wave::any(ft.serves && 8u * c < ft_len); c += tsz) {
                        if (ft.serves && 8u * c < ft_len) L.stage[(ft_stage >> 3) + c] = load_u64u(job.out + ft_src + 8u * c);
                    }
                }
            }
            wave::sync();
            clk.lap(kPhLvLong);

            // -- 5b. LZ77 copies in dependency levels.  A piece runs as soon as none of the pieces its
            //        source overlaps is still unfinished (dep_mask).  A level without long pieces runs one lane
            //        per piece; otherwise the ready pieces share the 32 lanes as teams, 8 bytes per lane per step.
            //        Overlapping copies replay their pattern modulo the distance, so a copy never waits for itself.
            {
                const uint32_t packed = plen | (far_len << 11) | ((stage_off >> 3) << 22);
                // simple piece: pattern in one place (window or staging area) and no chunk of a 32-byte batch reads
                // what an earlier chunk of the batch wrote
                const bool simple = (far_len == 0u || far_len == pattern) && (dist >= 32u || dist >= plen);
                uint32_t todo = wave::half_ballot(plen != 0u && !far_direct && !(kAblate & kAblLevels));
                while (wave::any(todo != 0u)) {
                    clk.count(kPhLevels, 1);
                    clk.halves(kPhLevelHalves, todo != 0u);
                    const bool ready = ((todo >> sl) & 1u) != 0u && (todo & dep_mask) == 0u;
                    const uint32_t ready_mask = wave::half_ballot(ready);
                    if ((kAblate & kAblTeams) || !wave::any(ready && (plen > (simple ? kOwnCopy : kShortCopy) || ((kAblate & kExpNoB) && !simple)))) {
                        // Own-lane copies.  The usual piece (pattern in one place; distance >= 32 or no overlap
                        // with itself) moves in batches of four 8-byte chunks, loads before stores, at offsets
                        // clipped to plen - 8: within a batch no chunk reads what an earlier chunk of the batch
                        // wrote, and every byte loaded belongs to the source (a piece ready in this level never
                        // has another ready piece inside its source).
                        const uint8_t* sp = far_len ? reinterpret_cast<const uint8_t*>(L.stage) + stage_off : L.win + (int32_t)src_idx;
                        uint8_t* dp = L.win + dst_idx;
                        const bool whole = far_len == 0u || far_len == pattern;
                        const bool lane_a = ready && simple && !(kAblate & kAblOwnLane);
                        const bool lane_b = ready && !simple && !(kAblate & (kAblOverlap | kExpNoB));
                        if (lane_a) {
                            if (plen >= 8u) {
                                const uint32_t c1 = min_u32(8u, clip8), c2 = min_u32(16u, clip8), c3 = min_u32(24u, clip8);
                                uint64_t v0, v1 = 0, v2 = 0, v3 = 0;
                                v0 = load_u64u(sp);
                                if (plen > 8u) v1 = load_u64u(sp + c1);
                                if (plen > 16u) v2 = load_u64u(sp + c2);
                                if (plen > 24u) v3 = load_u64u(sp + c3);
                                __builtin_memcpy(dp, &v0, 8);
                                if (plen > 8u) __builtin_memcpy(dp + c1, &v1, 8);
                                if (plen > 16u) __builtin_memcpy(dp + c2, &v2, 8);
                                if (plen > 24u) __builtin_memcpy(dp + c3, &v3, 8);
                            } else {
                                store_bytes(dp, load_u64u(sp), plen);
                            }
                        }
                        for (uint32_t o = 32u; wave::any(lane_a && plen > o); o += 32u) {      // further batches: bytes o .. min(o + 32, plen) - 1
                            if (lane_a && plen > o) {
                                const uint32_t c0 = min_u32(o, clip8), c1 = min_u32(o + 8u, clip8), c2 = min_u32(o + 16u, clip8), c3 = min_u32(o + 24u, clip8);
                                uint64_t v0, v1 = 0, v2 = 0, v3 = 0;
                                v0 = load_u64u(sp + c0);
                                if (plen > o + 8u) v1 = load_u64u(sp + c1);
                                if (plen > o + 16u) v2 = load_u64u(sp + c2);
                                if (plen > o + 24u) v3 = load_u64u(sp + c3);
                                __builtin_memcpy(dp + c0, &v0, 8);
                                if (plen > o + 8u) __builtin_memcpy(dp + c1, &v1, 8);
                                if (plen > o + 16u) __builtin_memcpy(dp + c2, &v2, 8);
                                if (plen > o + 24u) __builtin_memcpy(dp + c3, &v3, 8);
                            }
                        }
                        clk.lap(kPhLvShort);
                        if (wave::any(lane_b)) {
                            // The rest.  Self-overlapping pieces with a distance below 32 are copied forward in
                            // 8-byte chunks from `dd` bytes back, each chunk reading what its predecessors wrote
                            // (LDS accesses of a wave execute in order); a distance below 8 first lays down eight
                            // bytes of its pattern and then continues from the smallest multiple of itself that is
                            // >= 8 (8 - dd >= -dist: the read never reaches below the pattern).  Patterns that
                            // straddle the window boundary go byte by byte.
                            const uint8_t* own_stage = reinterpret_cast<const uint8_t*>(L.stage) + stage_off;
                            const uint8_t* own_win = L.win + (int32_t)src_idx;
                            uint32_t dd = dist, o0 = 0u, r = 0u;
                            if (lane_b && whole && dist < 8u) {
                                store_bytes(dp, pattern_source8(sp, dist, 0u), plen);
                                dd = (uint32_t)(0x0E0C0A0809080800ull >> (8u * dist)) & 0xFFu;     // 8, 8, 9, 8, 10, 12, 14 for 1..7
                                o0 = 8u;
                            }
                            for (uint32_t o = o0; wave::any(lane_b && o < plen); o += 8u) {
                                if (lane_b && o < plen) {
                                    uint64_t v;
                                    if (whole) v = load_u64u(dp + o - dd);
                                    else {
                                        v = 0;
                                        uint32_t rr = r;
                                        for (uint32_t b = 0; b < 8u; ++b) {
                                            const uint64_t x = rr < far_len ? own_stage[rr] : own_win[rr];
                                            v |= x << (8u * b);
                                            rr = rr + 1u == dist ? 0u : rr + 1u;
                                        }
                                        r = advance_mod(r, 8u, dist);
                                    }
                                    store_bytes(dp + o, v, plen - o);
                                }
                            }
                            clk.lap(kPhLvOverlap);
                        }
                    } else {
                    clk.count(kPhTeamLevels, 1);
                    const Team t = make_team(ready_mask, sl);
                    const uint32_t t_pk = wave::half_shfl(packed, t.job), t_dist = wave::half_shfl(dist, t.job);
                    const uint32_t t_src = wave::half_shfl(src_idx, t.job), t_dst = wave::half_shfl(dst_idx, t.job);
                    const uint32_t t_len = t_pk & 0x7FFu, t_far = (t_pk >> 11) & 0x7FFu;
                    const uint8_t* t_stage = reinterpret_cast<const uint8_t*>(L.stage) + ((t_pk >> 22) << 3);
                    const uint8_t* t_win = L.win + (int32_t)t_src;
                    uint8_t* t_out = L.win + t_dst;
                    const bool act = t.serves && ready_mask != 0u;
                    const uint32_t t_pat = t_dist < t_len ? t_dist : t_len;
                    const bool whole = t_far == 0u || t_far == t_pat;    // pattern in one place (window or staging area)
                    const uint8_t* t_base = t_far ? t_stage : t_win;
                    const bool overlap = t_dist < t_len;
                    clk.lap(kPhLvShort);
                    for (uint32_t c = t.member; wave::any(act && 8u * c < t_len); c += 1u << t.log2_size) {
                        const uint32_t j = 8u * c;
                        if (act && j < t_len) {
                            uint32_t r = j;
                            if (overlap) r = mod_u16(j, t_dist);
                            uint64_t v;
                            if (whole) v = pattern_source8(t_base, t_dist, r);
                            else {                                      // pattern straddles the window boundary: byte by byte
                                v = 0;
                                uint32_t rr = r;
                                for (uint32_t b = 0; b < 8u; ++b) {
                                    const uint32_t x = rr < t_far ? t_stage[rr] : t_win[rr];
                                    v |= (uint64_t)x << (8u * b);
                                    rr = rr + 1u == t_dist ? 0u : rr + 1u;
                                }
                            }
                            store_bytes(t_out + j, v, t_len - j);
                        }
                        wave::sync();
                    }
                    clk.lap(kPhLvBytes);
                    }
                    todo &= ~ready_mask;
                    wave::sync();
                }
            }
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

        // per-page delta decode of the colour sub-streams (PageDecoder.cpp:446-471), as in the fused kernel
        const bool do_delta = ended && is_delta != 0u;
        if (wave::any(do_delta)) {
            wave::global_fence();
            for (uint32_t c = 0; c < kMaxSubBlocks; ++c) {
                uint32_t lo = 0, hi = 0;
                if (do_delta && ((job.dc->color_mask >> c) & 1u)) {
                    const uint32_t cs = job.dc->sub_stream_off[c], ce = job.dc->sub_stream_off[c + 1];
                    const uint32_t ps = job.page_off, pe = job.page_off + job.out_size;
                    if (cs < pe && ps < ce) { lo = (cs > ps ? cs : ps) - ps; hi = (ce < pe ? ce : pe) - ps; }
                }
                uint32_t carry = 0;
                for (uint32_t base = lo & ~15u; wave::any(base < hi); base += 512u) {
                    const uint32_t pos = base + sl * 16u;
                    const bool full = pos >= lo && pos + 16u <= hi;
                    uint32_t w[4] = {0u, 0u, 0u, 0u};
                    if (full) {
                        __builtin_memcpy(w, __builtin_assume_aligned(job.out + pos, 16), 16);
                    } else {
                        for (uint32_t i = 0; i < 16u; ++i)
                            if (pos + i >= lo && pos + i < hi) w[i >> 2] |= (uint32_t)job.out[pos + i] << (8u * (i & 3u));
                    }
                    w[0] = byte_prefix(w[0]);
                    w[1] = byte_add(byte_prefix(w[1]), w[0] >> 24);
                    w[2] = byte_add(byte_prefix(w[2]), w[1] >> 24);
                    w[3] = byte_add(byte_prefix(w[3]), w[2] >> 24);
                    const uint32_t total = w[3] >> 24;
                    const uint32_t incl = wave::half_scan_incl(total) & 0xFFu;
                    const uint32_t add = (carry + incl - total) & 0xFFu;
                    for (uint32_t k = 0; k < 4u; ++k) w[k] = byte_add(w[k], add);
                    if (full) {
                        __builtin_memcpy(__builtin_assume_aligned(job.out + pos, 16), w, 16);
                    } else {
                        for (uint32_t i = 0; i < 16u; ++i)
                            if (pos + i >= lo && pos + i < hi) job.out[pos + i] = (uint8_t)(w[i >> 2] >> (8u * (i & 3u)));
                    }
                    carry = (carry + wave::half_shfl(incl, 31u)) & 0xFFu;
                }
            }
        }
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

        // per-page delta decode of the colour sub-streams (PageDecoder.cpp:446-471): lanes 0..31, as in the fused kernel
        if ((flags & kSlotDelta) != 0u) {
            wave::global_fence();
            const uint32_t sl = lane & 31u;
            const bool lower = lane < 32u;
            for (uint32_t c = 0; c < kMaxSubBlocks; ++c) {
                uint32_t lo = 0, hi = 0;
                if ((job.dc->color_mask >> c) & 1u) {
                    const uint32_t cs = job.dc->sub_stream_off[c], ce = job.dc->sub_stream_off[c + 1];
                    const uint32_t ps = job.page_off, pe = job.page_off + job.out_size;
                    if (cs < pe && ps < ce) { lo = (cs > ps ? cs : ps) - ps; hi = (ce < pe ? ce : pe) - ps; }
                }
                uint32_t carry = 0;
                for (uint32_t base = lo & ~15u; base < hi; base += 512u) {
                    const uint32_t pos = base + sl * 16u;
                    const bool full = lower && pos >= lo && pos + 16u <= hi;
                    uint32_t w[4] = {0u, 0u, 0u, 0u};
                    if (full) {
                        __builtin_memcpy(w, __builtin_assume_aligned(out + pos, 16), 16);
                    } else if (lower) {
                        for (uint32_t i = 0; i < 16u; ++i)
                            if (pos + i >= lo && pos + i < hi) w[i >> 2] |= (uint32_t)out[pos + i] << (8u * (i & 3u));
                    }
                    w[0] = byte_prefix(w[0]);
                    w[1] = byte_add(byte_prefix(w[1]), w[0] >> 24);
                    w[2] = byte_add(byte_prefix(w[2]), w[1] >> 24);
                    w[3] = byte_add(byte_prefix(w[3]), w[2] >> 24);
                    const uint32_t tot = w[3] >> 24;
                    const uint32_t incl = wave::half_scan_incl(tot) & 0xFFu;
                    const uint32_t add = (carry + incl - tot) & 0xFFu;
                    for (uint32_t k = 0; k < 4u; ++k) w[k] = byte_add(w[k], add);
                    if (full) {
                        __builtin_memcpy(__builtin_assume_aligned(out + pos, 16), w, 16);
                    } else if (lower) {
                        for (uint32_t i = 0; i < 16u; ++i)
                            if (pos + i >= lo && pos + i < hi) out[pos + i] = (uint8_t)(w[i >> 2] >> (8u * (i & 3u)));
                    }
                    carry = (carry + wave::half_shfl(incl, 31u)) & 0xFFu;
                }
            }
        }
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

__global__ void __launch_bounds__(64, BROTLIG_G_WAVES) brotlig_assemble_global_kernel(DecodeArgs a)
{
    __shared__ GlobalAsmLds L;
    assemble_pages_global(L, a);
}

}  // namespace brotlig
