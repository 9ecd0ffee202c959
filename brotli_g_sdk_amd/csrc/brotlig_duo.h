// brotlig_duo.h -- small batches: two wavefronts per page (brotlig_decode_duo_kernel).
// Part of the gfx950 Brotli-G decode kernels; brotlig_kernels.h includes the parts in order and says what the whole replaces.
#pragma once
#include "brotlig_round.h"

namespace brotlig {

// ===========================================================================================
// Small batches, second form: TWO wavefronts per page (brotlig_decode_duo_kernel).  With fewer pages than SIMDs a page's
// latency is the whole launch, and the latency of a page is its rounds times the dependent chain of one round.  That
// chain splits where the reference shader's phases split (BrotliGCompute.hlsl:761-1347 entropy, :1401-1419 assembly): the
// first wavefront of a workgroup decodes commands, distances and literals (everything that touches the bit streams), the
// second one assembles the output window from them (everything that touches the output), one group behind -- through a
// ring of kDuoSlots step records in LDS.  A step is one group of a round: its literals in consumption order and, for a
// round's first group, the round's 32 commands.  Page start / page end / "no more pages" travel through the same ring, so
// the two wavefronts never meet at a barrier: the producer is up to kDuoSlots - 1 steps ahead and builds the next page's
// tables while the consumer still flushes the last one.
// Hand-over: the producer fills a slot, then publishes `produced` (LDS store with release semantics, wave_ops.h); the
// consumer polls it, reads the slot, and gives it back through `consumed` as soon as the group's literals are in the
// window.
constexpr uint32_t kDuoSlots = 4;
// When the producer is this many steps ahead of the consumer (0 = never) it also does the group's dependency analysis -- which needs
// positions only -- and sends the masks along: on copy-dense pages the consumer is the longer half (samples16: 64 % of the fused time).
#ifndef BROTLIG_TUNE_DUO_DEPS_AHEAD
#define BROTLIG_TUNE_DUO_DEPS_AHEAD 1     // round 5, timed on the device: one samples16 page 0.688 -> 0.655 ms, records 0.343 -> 0.335, runs 0.277 -> 0.270, text even
#endif
enum : uint32_t { kDuoGroup = 1u, kDuoRound = 2u, kDuoPageStart = 4u, kDuoPageEnd = 8u, kDuoFinish = 16u, kDuoBad = 32u, kDuoDelta = 64u, kDuoDeps = 128u };
enum : uint32_t { kDuoOk = 1u << 31, kDuoCopies = 1u << 30 };         // flags above the distance (< 2^18)
struct __attribute__((aligned(16))) DuoStep {
    uint32_t kind, round_bytes, litcount, f0;                           // what the step is; sizes of its round; first literal of its group
    uint32_t ins[32], tot[32], dist[32], rel0[32], lit_a[32];           // the round's commands (steps with kDuoRound)
    uint32_t dep[32];                                                   // the group's dependency masks (steps with kDuoDeps)
    uint64_t lits[GeoSolo::kStageBytes / 8];                            // the group's literals, consumption order (or the PageJob of a page start)
};
static_assert(sizeof(PageJob) <= GeoSolo::kStageBytes, "a page start carries its job in the literal area");
typedef Geometry<256, 272, 720> GeoDuoEntropy;                          // the producer's record only needs the table-build scratch of stage / win
static_assert(GeoDuoEntropy::kWin + 16u >= kIcpAlphabet && sizeof(uint16_t) * (1 << kLutBitsLit) + GeoDuoEntropy::kStageBytes >= kTableScratchBytes,
              "table-build scratch of the producer's record");
struct __attribute__((aligned(16))) DuoLds {
    PageLdsT<GeoDuoEntropy> e;                                          // producer: tables, carry ring, distance ring
    uint64_t stage[GeoSolo::kStageBytes / 8];                           // consumer: far sources, piece bitmaps, output window
    uint32_t start_bits[GeoSolo::kRoundMax / 32];
    uint8_t  start_cum[GeoSolo::kRoundMax / 32];
    uint8_t  win[GeoSolo::kWin + 16] __attribute__((aligned(16)));
    DuoStep  step[kDuoSlots];
    uint32_t produced, consumed;                                        // steps handed over / given back so far
    uint32_t p_start_bits[GeoSolo::kRoundMax / 32];                     // the producer's own piece bitmaps (kDuoDeps)
    uint8_t  p_start_cum[GeoSolo::kRoundMax / 32];
    uint32_t len_code_tab[48];
};

// First wavefront: phases K4..K8 of the reference shader for one page at a time (its lower half; the upper half idles).  The
// code is decode_pages<> without steps 3b-3d, 4b and 5: see there for the comments on each stage.
__device__ inline void duo_producer(DuoLds& D, const DecodeArgs& a)
{
    typedef GeoSolo G;                                                  // group size of the consumer's window
    PhaseClock<false> clk;
    const uint32_t lane = wave::lane_id(), sl = lane & 31u;
    PageLdsT<GeoDuoEntropy>& L = D.e;
    uint16_t* const far_syms = a.far_syms + (size_t)blockIdx.x * (2u * kFarSymStride);
    const TableRef t_icp{L.lut_icp, L.sorted_icp, L.limit[0], L.first_offs[0], kIcpAlphabet, kLutBitsIcp, far_syms};
    const TableRef t_dist{L.lut_dist, L.sorted_dist, L.limit[1], L.first_offs[1], kDistAlphabet, kLutBitsDist, far_syms};
    const TableRef t_lit{L.lut_lit, L.sorted_lit, L.limit[2], L.first_offs[2], kLitAlphabet, kLutBitsLit, nullptr};

    PageJob job = no_job(a);
    bool live = false, finished = lane >= 32u, bad = false;
    BitReader br;
    br.base = a.in; br.limit8 = 0; br.buf = 0; br.avail = 64; br.next = 0; br.queue = 0; br.queued = 64; br.flight = 0; br.zero = wave::opaque_zero(); br.slot = nullptr; br.slot0 = 0;
    DistanceRing ring;
    uint32_t out_pos = 0, prev_tail = 0, carry_head = 0;
    uint32_t k = 0;                                                     // steps produced so far
    auto acquire = [&D](uint32_t step) -> DuoStep& {
        while (step - wave::lds_load_acquire(&D.consumed) >= kDuoSlots) wave::nap();
        return D.step[step % kDuoSlots];
    };
    auto publish = [&D, lane](uint32_t steps) { wave::sync(); if (lane == 0u) wave::lds_store_release(&D.produced, steps); };

    for (;;) {
        if (wave::any(!live && !finished)) {
            bool tables_ok = true;
            const bool start = start_pages(a, L, job, br, !live && !finished, finished, sl, far_syms, tables_ok, [](const PageJob&) {}, clk);
            ring.reset(L, start, sl);
            if (start) {
                out_pos = 0; prev_tail = 0; carry_head = 0; bad = false;
                live = true;
                if (!tables_ok) { bad = true; out_pos = job.out_size; }  // the first round is refused (or is a bare sentinel): nothing is assembled
            }
        }
        if (!wave::any(live)) break;
        {
            DuoStep& S = acquire(k);
            if (lane == 0u) { S.kind = kDuoPageStart; *reinterpret_cast<PageJob*>(S.lits) = job; }
            publish(++k);
        }
        const bool in_page = live;
        do {
            const RingWords pushed = load_ring_pushes(L, ring);
            RoundCommands cmd = decode_round_commands(L, D.len_code_tab, t_icp, t_dist, br, live, sl, clk);
            const uint32_t sent_mask = cmd.sent_mask, n = cmd.n;
            const bool is_cmd = cmd.is_cmd;
            resolve_distance_ring(L, ring, pushed, cmd, sl);
            const uint32_t ins = cmd.ins, copy = cmd.copy, dist = cmd.dist;
            const uint32_t tot = ins + copy;
            const uint32_t incl_tot = wave::half_scan_incl(tot);
            const uint32_t incl_ins = wave::half_scan_incl(ins);
            const uint32_t round_bytes = wave::half_bcast(incl_tot, 31);
            const uint32_t litcount = wave::half_bcast(incl_ins, 31);
            const uint32_t copy_dst = out_pos + incl_tot - tot + ins;
            if (live && round_bytes > job.out_size - out_pos) { bad = true; live = false; }
            const bool ok_cmd = is_cmd && live;
            const uint32_t lit_a = incl_ins - ins, rel0 = incl_tot - tot;
            const uint32_t ac = litcount > prev_tail ? litcount - prev_tail : 0u;
            const uint32_t mult = (live && n) ? div_small(min_u32(ac, 0x200000u) + n - 1u, n) : 0u;
            const uint32_t rlit = n * mult;
            uint32_t next_j = sl;
            const bool dist_ok = dist != 0u && dist <= copy_dst;
            if (ok_cmd && copy > 0u && !dist_ok) bad = true;
            const bool cp = ok_cmd && copy > 0u && dist_ok;

            const uint32_t ngroups = live ? (round_bytes + G::kRoundMax - 1u) / G::kRoundMax : 0u;
            const bool multi_group = wave::any(ngroups > 1u);
            for (uint32_t g = 0; wave::any(g < ngroups); ++g) {
                const bool on = g < ngroups;
                const uint32_t g0 = g * G::kRoundMax, g1 = on ? min_u32(round_bytes, g0 + G::kRoundMax) : g0;
                const uint32_t cs = rel0 + ins;
                const bool in_group = on && ok_cmd && rel0 < g1 && rel0 + tot > g0;
                const uint32_t la = rel0 > g0 ? rel0 : g0, lb = cs < g1 ? cs : g1;
                const uint32_t nlit = (in_group && lb > la) ? lb - la : 0u;
                const uint32_t mine_before = (on && ok_cmd) ? (cs <= g0 ? ins : (rel0 < g0 ? g0 - rel0 : 0u)) : 0u;
                uint32_t F0 = 0, F1 = litcount;
                if (multi_group) { F0 = wave::half_sum(mine_before); F1 = F0 + wave::half_sum(nlit); }

                DuoStep& S = acquire(k);
                if (g == 0u && lane < 32u) {
                    S.ins[sl] = ok_cmd ? ins : 0u; S.tot[sl] = ok_cmd ? tot : 0u; S.rel0[sl] = rel0; S.lit_a[sl] = lit_a;
                    // (the distance only where the copy is valid: a damaged stream's ring code can wrap below zero -- 1 - 3 -- and its high bits
                    // would read as the flags; a copy that is not valid is not made, as in decode_pages)
                    S.dist[sl] = (cp ? dist : 0u) | (ok_cmd ? kDuoOk : 0u) | (cp ? kDuoCopies : 0u);
                }
                uint32_t with_deps = 0u;
                if (BROTLIG_TUNE_DUO_DEPS_AHEAD != 0) {
                    const uint32_t ahead = wave::bcast(k - wave::lds_load_acquire(&D.consumed), 0u);
                    if (ahead >= (uint32_t)BROTLIG_TUNE_DUO_DEPS_AHEAD) {
                        const uint32_t ca = cs > g0 ? cs : g0, cb = rel0 + tot < g1 ? rel0 + tot : g1;
                        const uint32_t plen = (in_group && cp && cb > ca) ? cb - ca : 0u;
                        const uint32_t pdst = out_pos + ca, psrc = pdst - (cp ? dist : 0u);
                        const uint32_t src_end = psrc + min_u32(plen, dist);
                        const uint32_t dep = piece_dependencies<PhaseClock<false>, G>(D.p_start_bits, D.p_start_cum, on, wave::ballot64(in_group), la - g0, out_pos + g0,
                                                                                     psrc, src_end, plen != 0u, sl, clk);
                        if (lane < 32u) S.dep[sl] = dep;
                        with_deps = kDuoDeps;
                    }
                }
                if (lane == 0u) { S.kind = kDuoGroup | (g == 0u ? kDuoRound : 0u) | with_deps; S.round_bytes = round_bytes; S.litcount = litcount; S.f0 = F0; }
                uint8_t* const lits = reinterpret_cast<uint8_t*>(S.lits);
                if (on) {
                    const uint32_t cf1 = F1 < prev_tail ? F1 : prev_tail;
                    for (uint32_t f = F0 + sl; f < cf1; f += 32u) lits[f - F0] = L.carry[(carry_head + f) & 63u];
                    const bool last_group = g + 1u == ngroups;
                    const uint32_t J1 = last_group ? rlit : (F1 > prev_tail ? F1 - prev_tail : 0u);
                    const uint32_t keep_at = carry_head + prev_tail;
                    auto place = [&](uint32_t j, uint32_t lit) {
                        const uint32_t f = prev_tail + j;
                        if (f < litcount) lits[f - F0] = (uint8_t)lit;
                        else L.carry[(keep_at + (f - litcount)) & 63u] = (uint8_t)lit;
                    };
                    for (; next_j + 32u < J1; next_j += 64u) {
                        uint32_t l0, l1;
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
                publish(++k);
            }
            if (live) {
                carry_head += min_u32(prev_tail, litcount);
                prev_tail = rlit + prev_tail - litcount;
                out_pos += round_bytes;
            }
            if (sent_mask) live = false;
        } while (!wave::any(in_page && !live));

        const bool ended = in_page && !live;
        if (ended && out_pos != job.out_size) bad = true;
        {
            const uint64_t verdict = wave::ballot64(ended && bad);
            DuoStep& S = acquire(k);
            if (lane == 0u) S.kind = kDuoPageEnd | (verdict != 0ull ? kDuoBad : 0u) | ((L.page_params >> 16) != 0u ? kDuoDelta : 0u);
            publish(++k);
        }
        if (ended && bad && sl == 0u) flag_bad_page(a, L.page_stream);
    }
    DuoStep& S = acquire(k);
    if (lane == 0u) S.kind = kDuoFinish;
    publish(++k);
}

// Second wavefront: phase K9 (LZ77 assembly, BrotliGCompute.hlsl:1401-1419; PageDecoder.cpp:209-233) from the step records, with
// the window, staging area, dependency levels and copy teams of decode_pages<> (steps 3b-3d, 4b, 5a, 5b there).  The page
// lives in the lower half; the upper half joins the copy teams of long pieces (kSolo in copy_levels).
__device__ inline void duo_consumer(DuoLds& D, const DecodeArgs& a)
{
    typedef GeoSolo G;
    PhaseClock<false> clk;
    const uint32_t lane = wave::lane_id(), sl = lane & 31u;
    const bool lower = lane < 32u;
    PageJob job = no_job(a);
    OutView view{D.win, 0u};
    uint32_t out_pos = 0, flushed = 0;
    uint32_t k = 0;                                                     // steps consumed so far
    auto wait_for = [&D](uint32_t step) -> DuoStep& {
        while (wave::lds_load_acquire(&D.produced) <= step) wave::nap();
        return D.step[step % kDuoSlots];
    };
    auto give_back = [&D, lane](uint32_t steps) { wave::sync(); if (lane == 0u) wave::lds_store_release(&D.consumed, steps); };

    for (;;) {
        DuoStep* S = &wait_for(k);
        const uint32_t kind = wave::uniform(S->kind);
        if (kind & kDuoFinish) break;
        if (kind & kDuoPageStart) {
            job = *reinterpret_cast<const PageJob*>(S->lits);
            out_pos = 0; flushed = 0; view.win_base = 0u;
            give_back(++k);
            continue;
        }
        if (kind & kDuoPageEnd) {
            wave::sync();
            if (lower) flushed = flush_window(job.out, view, flushed, out_pos, true, sl);
            delta_decode_page(job, lower && (kind & kDuoDelta) != 0u && (kind & kDuoBad) == 0u, sl);
            give_back(++k);
            continue;
        }
        // a round: its commands come with its first group
        const uint32_t ins = lower ? S->ins[sl] : 0u, tot = lower ? S->tot[sl] : 0u, rel0 = lower ? S->rel0[sl] : 0u, lit_a = lower ? S->lit_a[sl] : 0u;
        const uint32_t dword = lower ? S->dist[sl] : 0u;
        const uint32_t dist = dword & 0x3FFFFFFFu;
        const bool ok_cmd = (dword & kDuoOk) != 0u, cp = (dword & kDuoCopies) != 0u;
        const uint32_t round_bytes = wave::uniform(S->round_bytes), litcount = wave::uniform(S->litcount);
        const uint32_t ngroups = (round_bytes + G::kRoundMax - 1u) / G::kRoundMax;
        for (uint32_t g = 0; g < ngroups; ++g) {
            if (g != 0u) S = &wait_for(k);
            const bool on = lower;
            const uint32_t F0 = wave::uniform(S->f0);
            const uint32_t g0 = g * G::kRoundMax, g1 = min_u32(round_bytes, g0 + G::kRoundMax);
            const uint32_t gpos = out_pos + g0;

            flush_and_slide<G>(view, flushed, job.out, on, gpos, out_pos + g1, sl);
            const uint32_t span0 = gpos - view.win_base;

            const uint32_t cs = rel0 + ins;
            const bool in_group = on && ok_cmd && rel0 < g1 && rel0 + tot > g0;
            const uint32_t la = rel0 > g0 ? rel0 : g0, lb = cs < g1 ? cs : g1;
            const uint32_t nlit = (in_group && lb > la) ? lb - la : 0u;
            const uint32_t lit_f = lit_a + (la - rel0);
            const uint32_t ca = cs > g0 ? cs : g0, cb = rel0 + tot < g1 ? rel0 + tot : g1;
            const uint32_t plen = (in_group && cp && cb > ca) ? cb - ca : 0u;
            const uint32_t pdst = out_pos + ca;
            const uint32_t psrc = pdst - dist;
            const uint32_t pattern = min_u32(plen, dist);
            const uint32_t src_end = psrc + pattern;
            const uint32_t far_len = (plen && psrc < view.win_base) ? min_u32(pattern, view.win_base - psrc) : 0u;
            const FarSources far = fetch_far_sources(job.out, D.win, psrc - view.win_base, false, plen, psrc, far_len, sl);
            const bool far_direct = far.direct;
            const uint32_t stage_off = far.stage_off;
            uint32_t dep_mask;
            if (BROTLIG_TUNE_DUO_DEPS_AHEAD != 0 && (wave::uniform(S->kind) & kDuoDeps) != 0u) dep_mask = lower ? S->dep[sl] : 0u;
            else dep_mask = piece_dependencies<PhaseClock<false>, G>(D.start_bits, D.start_cum, on, wave::ballot64(in_group), (rel0 > g0 ? rel0 : g0) - g0, gpos,
                                                                              psrc, src_end, plen != 0u && !far_direct, sl, clk);
            // literal runs: from the step's queue to their place in the window; then the slot goes back to the producer
            {
                const uint8_t* const lits = reinterpret_cast<const uint8_t*>(S->lits);
                const uint32_t q_idx = lit_f - F0, w_idx = span0 - g0 + la;
                own_copy_simple(lits + q_idx, D.win + w_idx, nlit, wave::ballot_lt_k<kOwnCopy>(nlit - 1u));
                const uint64_t long_w = wave::ballot_gt_k<kOwnCopy>(nlit);
                if (long_w != 0ull) {
                    const uint32_t lmask = wave::half_of(long_w);
                    const Team tl = make_team(lmask, sl);
                    const uint32_t l_src = wave::half_shfl(q_idx, tl.job), l_dst = wave::half_shfl(w_idx, tl.job);
                    const uint32_t l_len = wave::half_shfl(nlit, tl.job);
                    const bool act = lower && tl.serves && lmask != 0u;
                    for (uint32_t c = tl.member; wave::any(act && 8u * c < l_len); c += 1u << tl.log2_size) {
                        const uint32_t j = 8u * c;
                        if (act && j < l_len) store_bytes(D.win + l_dst + j, load_u64u(lits + l_src + j), l_len - j);
                    }
                }
            }
            give_back(++k);

            const uint32_t src_idx = psrc - view.win_base, dst_idx = pdst - view.win_base;
            store_far_sources(D.win, D.stage, job.out, far, plen, far_len, dst_idx);
            wave::sync();
            {
                const uint64_t plain_w = (wave::ballot_eq0(far_len) | wave::ballot_eq(far_len, pattern)) & ~wave::ballot_lt(dist, plen) & wave::ballot_lt_k<33u>(plen);
                if ((wave::ballot_ne0(plen) & ~far.direct_w & ~plain_w) == 0ull)
                    copy_levels_plain(D.win, D.stage, plen, far_len, stage_off, src_idx, dst_idx, far.direct_w, dep_mask, sl, clk);
                else
                    copy_levels(D.win, D.stage, plen, dist, far_len, stage_off, src_idx, dst_idx, far.direct_w, dep_mask, sl, true, clk);
            }
        }
        out_pos += round_bytes;                                         // (a round without bytes sends no step)
        (void)litcount;
    }
}

__global__ void __launch_bounds__(128, 4) brotlig_decode_duo_kernel(DecodeArgs a)     // (4 wavefronts per SIMD = the 8 workgroups per compute unit its LDS allows: 128 registers -- round 5: the new table builder had taken 145)
{
    __shared__ DuoLds D;
    const uint32_t t = threadIdx.x;
    if (t < 48u) D.len_code_tab[t] = kLenCodeTab[t];
    if (t == 0u) { D.produced = 0u; D.consumed = 0u; }
    __syncthreads();
    const uint32_t total = a.page_base[a.num_streams];
    if (blockIdx.x >= total || total > a.duo_limit) return;             // more workgroups than pages, or a batch for brotlig_decode_kernel
    if (t < 64u) duo_producer(D, a); else duo_consumer(D, a);
}

}  // namespace brotlig
