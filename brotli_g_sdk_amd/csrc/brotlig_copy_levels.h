// brotlig_copy_levels.h -- assembly stages of a group: window flush / slide, far sources, piece dependencies, LZ77 copies in dependency levels, per-page delta (PageDecoder.cpp:209-233, :446-471).
// Part of the gfx950 Brotli-G decode kernels; brotlig_kernels.h includes the parts in order and says what the whole replaces.
#pragma once
#include "brotlig_kernel_common.h"
#include "brotlig_jobs.h"

namespace brotlig {

// ---- stage: flush and slide of the output window, at the start of a group whose first byte is page position `gpos` and
// whose last is `gend - 1`.  Every group first stores the finished bytes below it (aligned 16-byte pieces; `flushed` is
// 16-byte aligned until the page's last flush and at most kRoundMax + 15 bytes behind), so that a far copy -- source
// below the window, i.e. more than kHist >= kRoundMax + 16 bytes back -- only ever reads global memory written by an
// EARLIER group's flush.  When the group does not fit behind what the window holds, the window slides: kHist .. kHist + 15
// bytes of history are kept and brought down in one step, all reads before the writes.
template <class G = GeoPair>
__device__ __forceinline__ void flush_and_slide(OutView& view, uint32_t& flushed, uint8_t* out, bool on, uint64_t on_w, uint32_t gpos, uint32_t gend, uint32_t sl)
{
    const uint64_t slide_w = (kAblate & kAblSlide) ? 0ull : on_w & wave::ballot_gt(gend, view.win_base + G::kWin);       // (on_w: `on` of every lane)
    const bool slide = wave::from_mask(slide_w);
    wave::sync();
    // (the two-piece and the three-piece forms are written out separately: with the third piece as a folded-away branch inside
    // the two-piece code the compiler dropped the skip branches around the second store and issued it with an empty mask --
    // one global store and one load more per round, 4.7 % on the mixed data; round 4)
    if constexpr (G::kFlushPieces <= 2u) {
        const uint32_t e16 = gpos & ~15u;
        const uint32_t p0 = flushed + 16u * sl, p1 = p0 + 512u;
        const bool f0 = on && p0 < e16, f1 = on && p1 < e16;
        Bytes16 a0, a1;         // (each stored under the condition it is loaded under: zeroing them is four v_mov apiece)
        if (f0) a0 = load16(view.win + (p0 - view.win_base));
        if (f1) a1 = load16(view.win + (p1 - view.win_base));
        if (f0) store16(out + p0, a0);
        if (f1) store16(out + p1, a1);
        if (on && e16 > flushed) flushed = e16;
    } else {
        const uint32_t e16 = gpos & ~15u;
        const uint32_t p0 = flushed + 16u * sl, p1 = p0 + 512u, p2 = p0 + 1024u;
        const bool f0 = on && p0 < e16, f1 = on && p1 < e16, f2 = on && p2 < e16;
        Bytes16 a0, a1, a2;
        if (f0) a0 = load16(view.win + (p0 - view.win_base));
        if (f1) a1 = load16(view.win + (p1 - view.win_base));
        if (f2) a2 = load16(view.win + (p2 - view.win_base));
        if (f0) store16(out + p0, a0);
        if (f1) store16(out + p1, a1);
        if (f2) store16(out + p2, a2);
        if (on && e16 > flushed) flushed = e16;
    }
    if (slide_w != 0ull) {
        const uint32_t nb = slide ? (gpos - G::kHist) & ~15u : view.win_base;
        const uint32_t shift = nb - view.win_base, count = shift ? gpos - nb : 0u;
        if constexpr (G::kSlidePieces <= 2u) {
            const uint32_t i0 = 16u * sl, i1 = 512u + 16u * sl;
            Bytes16 m0, m1;
            if (i0 < count) m0 = load16(view.win + shift + i0);
            if (i1 < count) m1 = load16(view.win + shift + i1);
            wave::sync();
            if (i0 < count) store16(view.win + i0, m0);
            if (i1 < count) store16(view.win + i1, m1);
        } else {
            const uint32_t i0 = 16u * sl, i1 = 512u + 16u * sl, i2 = 1024u + 16u * sl;
            Bytes16 m0, m1, m2;
            if (i0 < count) m0 = load16(view.win + shift + i0);
            if (i1 < count) m1 = load16(view.win + shift + i1);
            if (i2 < count) m2 = load16(view.win + shift + i2);
            wave::sync();
            if (i0 < count) store16(view.win + i0, m0);
            if (i1 < count) store16(view.win + i1, m1);
            if (i2 < count) store16(view.win + i2, m2);
        }
        view.win_base = nb;
    }
    wave::sync();
}
template <class G = GeoPair>
__device__ __forceinline__ void flush_and_slide(OutView& view, uint32_t& flushed, uint8_t* out, bool on, uint32_t gpos, uint32_t gend, uint32_t sl)
{
    flush_and_slide<G>(view, flushed, out, on, wave::ballot64(on), gpos, gend, sl);
}

// ---- stage: the LZ77 copies of a group in dependency levels (PageDecoder.cpp:219-232 / BrotliGCompute.hlsl:1401-1419).
// One piece per lane: `plen` bytes to window index dst_idx from `dist` bytes back; the first far_len bytes of its pattern
// come from the staging area at stage_off (far sources, stored there before the call), the rest from the window at src_idx.
// A piece runs as soon as none of the pieces its source overlaps is still unfinished (dep_mask: lanes of the half).  A
// level without long pieces runs one lane per piece; otherwise the ready pieces share the 32 lanes as teams, 8 bytes per
// lane per step.  Overlapping copies replay their pattern modulo the distance, so a copy never waits for itself.
// Pieces with `far_direct` went from registers straight to their place and take no part.
#ifndef BROTLIG_TUNE_POW2_OVERLAP
#define BROTLIG_TUNE_POW2_OVERLAP 1
#endif
#ifndef BROTLIG_TUNE_PLAIN_LEVELS
#define BROTLIG_TUNE_PLAIN_LEVELS 1
#endif
// Runs first (round 6): in a team level that holds runs (pieces that repeat a period of 1, 2, 4 or 8 bytes) AND other pieces, the runs are
// served first, by teams of their own and with the cheap loop (one pattern word per piece, read by its owner; no remainder and no pattern
// read per chunk); the general loop then serves the rest with more lanes per piece.  Until then a level was ONE pass over all ready pieces
// and took the cheap loop only when every served piece of BOTH halves was a run -- a quarter of the team levels of run-length pages, whose
// levels hold 2.1 runs and 0.7 other pieces on average (profiles/tools/cmd_stats.c).  Run-length pages +16 %, config 2 +7.5 %.  In the
// kernel as it was compiled until round 6 the code cost every other class 1 .. 2.5 % by its presence (register allocation of the round
// loop); built without the loop unroller's choices (-fno-unroll-loops, _build.HIP_FLAGS: forty spilled values fewer) it costs none.
// 0: the single pass.
#ifndef BROTLIG_TUNE_DELTA_UNROLL
#define BROTLIG_TUNE_DELTA_UNROLL 1
#endif
#ifndef BROTLIG_TUNE_RUNS_FIRST
#define BROTLIG_TUNE_RUNS_FIRST 1
#endif
// The same for a group in which every piece that takes part is simple and at most 32 bytes long (decided once per group by the
// caller, for both halves): a level is then one batch of own-lane chunk copies and nothing else -- no question about teams, about
// further batches or about the overlap path in any iteration.  (Round 4: those three questions are ~19 of a level's ~90 issued
// instructions, 3.7 levels a round; text pages take this path in nearly every group.)
template <class Clock>
__device__ __forceinline__ void copy_levels_plain(uint8_t* win, const uint64_t* stage, uint32_t plen, uint32_t far_len, uint32_t stage_off,
                                                  uint32_t src_idx, uint32_t dst_idx, uint64_t direct_w, uint32_t dep_mask, uint32_t sl, Clock& clk)
{
    const uint32_t clip8 = plen >= 8u ? plen - 8u : 0u;
    const uint8_t* const sp = far_len ? reinterpret_cast<const uint8_t*>(stage) + stage_off : win + (int32_t)src_idx;
    uint8_t* const dp = win + dst_idx;
    // The level loop's questions as wave-wide lane masks in scalar registers (wave::from_mask turns a mask back into a lane predicate
    // without an instruction): the ballot of a COMPOUND predicate goes through a 0 / 1 register and a second compare.
    uint64_t todo_w = wave::ballot_ne0(plen) & ~direct_w;
    uint32_t todo = wave::half_of(todo_w);
    const uint64_t ge8_w = wave::ballot_gt_k<7u>(plen);
    while (todo_w != 0ull) {
        clk.count(kPhLevels, 1);
        clk.halves(kPhLevelHalves, todo != 0u);
        const uint64_t ready_w = todo_w & wave::ballot_eq0(todo & dep_mask);
        if (wave::from_mask(ready_w & ge8_w)) {
            const Chunks32 c = load_chunks32(sp, plen, clip8);
            store_chunks32(dp, c, plen, clip8);
        }
        if (wave::from_mask(ready_w & ~ge8_w)) store_bytes(dp, load_u64u(sp), plen);
        clk.lap(kPhLvShort);
        todo &= ~wave::half_of(ready_w);
        todo_w &= ~ready_w;
        wave::sync();
    }
    (void)sl;
}

template <class Clock>
__device__ __forceinline__ void copy_levels(uint8_t* win, const uint64_t* stage, uint32_t plen, uint32_t dist, uint32_t far_len, uint32_t stage_off,
                                            uint32_t src_idx, uint32_t dst_idx, uint64_t direct_w, uint32_t dep_mask, uint32_t sl, bool solo, Clock& clk)
{
    const uint32_t pattern = min_u32(plen, dist);
    const uint32_t clip8 = plen >= 8u ? plen - 8u : 0u;
    const uint32_t packed = plen | (far_len << 11) | ((stage_off >> 3) << 22);
    // simple piece: pattern in one place (window or staging area) and no chunk of a 32-byte batch reads
    // what an earlier chunk of the batch wrote
    // (the questions of the level loop as wave-wide lane masks in scalar registers, see copy_levels_plain)
    uint64_t todo_w = (kAblate & kAblLevels) ? 0ull : wave::ballot_ne0(plen) & ~direct_w;
    uint32_t todo = wave::half_of(todo_w);
    const uint64_t whole_w = wave::ballot_eq0(far_len) | wave::ballot_eq(far_len, pattern);         // pattern in one place
    const uint64_t simple_w = whole_w & (wave::ballot_gt_k<31u>(dist) | ~wave::ballot_lt(dist, plen));
    const bool simple = wave::from_mask(simple_w);
    const uint64_t ge8_w = wave::ballot_gt_k<7u>(plen), gt32_w = wave::ballot_gt_k<32u>(plen);
#if BROTLIG_TUNE_POW2_OVERLAP
    // self-overlapping pieces with a period of 1, 2 or 4 bytes whose pattern lies in one place (asked once per group, and only when there is a
    // piece that overlaps itself at all)
    const uint64_t pow2_dist_w = (~simple_w & todo_w) != 0ull ? whole_w & ~simple_w & wave::ballot_lt_k<5u>(dist) & ~wave::ballot_eq_k<3u>(dist) : 0ull;
#else
    const uint64_t pow2_dist_w = 0ull;
#endif
    // long pieces make their level a team level: simple ones beyond kOwnCopy bytes, others beyond kShortCopy -- except the periods of 1, 2 and 4
    // bytes, which are one word stored kOverlapOwn / 8 times at most by their own lane (round 6: 48 instead of 32 -- 16-bit samples +3.7 %)
    const uint64_t long_w = (simple_w & wave::ballot_gt_k<kOwnCopy>(plen)) | (~simple_w & ~pow2_dist_w & wave::ballot_gt_k<kShortCopy>(plen)) |
                            (pow2_dist_w & wave::ballot_gt_k<kOverlapOwn>(plen));
    while (todo_w != 0ull) {
        clk.count(kPhLevels, 1);
        clk.halves(kPhLevelHalves, todo != 0u);
        const uint64_t ready_w = todo_w & wave::ballot_eq0(todo & dep_mask);
        const bool ready = wave::from_mask(ready_w);
        const uint32_t ready_mask = wave::half_of(ready_w);
        if ((kAblate & kAblTeams) || (ready_w & long_w) == 0ull) {
            // Own-lane copies.  The usual piece (pattern in one place; distance >= 32 or no overlap
            // with itself) moves in batches of four 8-byte chunks, loads before stores, at offsets
            // clipped to plen - 8: within a batch no chunk reads what an earlier chunk of the batch
            // wrote, and every byte loaded belongs to the source (a piece ready in this level never
            // has another ready piece inside its source).
            const uint8_t* sp = far_len ? reinterpret_cast<const uint8_t*>(stage) + stage_off : win + (int32_t)src_idx;
            uint8_t* dp = win + dst_idx;
            const bool whole = wave::from_mask(whole_w);
            const uint64_t a_w = (kAblate & kAblOwnLane) ? 0ull : ready_w & simple_w, b_w = (kAblate & kAblOverlap) ? 0ull : ready_w & ~simple_w;
            if (wave::from_mask(a_w & ge8_w)) {
                const Chunks32 c = load_chunks32(sp, plen, clip8);
                store_chunks32(dp, c, plen, clip8);
            }
            if (wave::from_mask(a_w & ~ge8_w)) store_bytes(dp, load_u64u(sp), plen);
            uint64_t more_w = a_w & gt32_w;
            for (uint32_t o = 32u; more_w != 0ull; o += 32u, more_w &= wave::ballot_gt(plen, o)) {      // further batches: bytes o .. min(o + 32, plen) - 1
                if (wave::from_mask(more_w)) {
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
#if BROTLIG_TUNE_POW2_OVERLAP
            // Periods of 1, 2 and 4 bytes (a repeated byte, 16-bit sample, 32-bit word: most self-overlapping pieces of sampled data; here
            // the piece is at most 32 bytes, longer ones run in teams): the pattern as ONE 8-byte word whose halves are alike, stored at 0 / 8 /
            // 16 and, rotated to its phase, at plen - 8.  Nothing the piece wrote is read back: no LDS round trip per chunk.
            uint64_t rest_w = b_w;
            {
                const uint64_t pow2_w = b_w & pow2_dist_w;
                if (pow2_w != 0ull) {
                    if (wave::from_mask(pow2_w)) {
                        // (everything here depends on an opaque zero: otherwise the loop-invariant part -- phase, addresses, comparisons -- is
                        // hoisted in front of the level loop, paid by every group and kept in registers across the levels: text -2 %)
                        const uint32_t z = wave::opaque_zero();
                        const uint32_t d = dist | z, n = plen | z, c8 = n - 8u;
                        uint8_t* const q = dp + z;
                        uint32_t x;
                        __builtin_memcpy(&x, sp + z, 4);
                        const uint32_t w = d == 4u ? x : d == 2u ? (x & 0xFFFFu) * 0x00010001u : (x & 0xFFu) * 0x01010101u;
                        const uint32_t ph = 8u * (c8 & (d - 1u));                  // phase of the chunk that ends the piece
                        const uint32_t wt = ph ? (w >> ph) | (w << (32u - ph)) : w;
                        const uint64_t v = (uint64_t)w | ((uint64_t)w << 32), vt = (uint64_t)wt | ((uint64_t)wt << 32);
                        if (n >= 8u) {
                            __builtin_memcpy(q, &v, 8);
                            if (n > 8u) __builtin_memcpy(q + c8, &vt, 8);
                            if (n >= 16u) __builtin_memcpy(q + 8u, &v, 8);
                            if (n >= 24u) __builtin_memcpy(q + 16u, &v, 8);
                            if constexpr (kOverlapOwn > 32u) {          // (whole words from 24 on: the last one is the rotated word at n - 8)
                                if (n > 32u) {
#pragma unroll
                                    for (uint32_t o = 24u; o + 8u < kOverlapOwn; o += 8u)
                                        if (n > o + 8u) __builtin_memcpy(q + o, &v, 8);
                                }
                            }
                        } else store_bytes(q, v, n);
                    }
                    rest_w &= ~pow2_w;
                }
            }
            const bool lane_b = wave::from_mask(rest_w);
            if (rest_w != 0ull) {
#else
            const bool lane_b = wave::from_mask(b_w);
            if (b_w != 0ull) {
#endif
                // The rest.  Self-overlapping pieces with a distance below 32 are copied forward in
                // 8-byte chunks from `dd` bytes back, each chunk reading what its predecessors wrote
                // (LDS accesses of a wave execute in order); a distance below 8 first lays down eight
                // bytes of its pattern and then continues from the smallest multiple of itself that is
                // >= 8 (8 - dd >= -dist: the read never reaches below the pattern).  Patterns that
                // straddle the window boundary go byte by byte.
                const uint8_t* own_stage = reinterpret_cast<const uint8_t*>(stage) + stage_off;
                const uint8_t* own_win = win + (int32_t)src_idx;
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
        uint32_t serve_mask = ready_mask;       // the pieces (lanes of the half) that the teams below serve
#if BROTLIG_TUNE_RUNS_FIRST
        {
            // runs: the piece overlaps itself with a period of 1, 2, 4 or 8 bytes and its pattern lies in one place (a piece that lies in one
            // place and is not simple has a distance below 32 and below its length).  Asked here and not once per group: a group's masks live
            // in scalar registers across the level loop, and the loop has none to spare.
            const uint64_t runs_w = ready_w & whole_w & ~simple_w & ((wave::ballot_lt_k<5u>(dist) & ~wave::ballot_eq_k<3u>(dist)) | wave::ballot_eq_k<8u>(dist));
            if (runs_w != 0ull && runs_w != ready_w) {
                const bool run = wave::from_mask(runs_w);
                uint64_t word = 0;                                  // the owner reads its pattern; the team members get the word
                if (run) word = pattern_source8(far_len ? reinterpret_cast<const uint8_t*>(stage) + stage_off : win + (int32_t)src_idx, dist, 0u);
                uint32_t r_mask = wave::half_of(runs_w);
                Team rt;
                uint32_t r_lo, r_hi, r_len, r_dst;
                if (solo) {
                    r_mask = wave::bcast(r_mask, 0u);
                    rt = make_team64(r_mask, wave::lane_id());
                    r_lo = wave::bcast((uint32_t)word, rt.job); r_hi = wave::bcast((uint32_t)(word >> 32), rt.job);
                    r_len = wave::bcast(plen, rt.job); r_dst = wave::bcast(dst_idx, rt.job);
                } else {
                    rt = make_team(r_mask, sl);
                    r_lo = wave::half_shfl((uint32_t)word, rt.job); r_hi = wave::half_shfl((uint32_t)(word >> 32), rt.job);
                    r_len = wave::half_shfl(plen, rt.job); r_dst = wave::half_shfl(dst_idx, rt.job);
                }
                const bool r_act = rt.serves && r_mask != 0u;
                const uint64_t v = (uint64_t)r_lo | ((uint64_t)r_hi << 32);
                for (uint32_t c = rt.member; wave::any(r_act && 8u * c < r_len); c += 1u << rt.log2_size) {
                    const uint32_t j = 8u * c;
                    if (r_act && j < r_len) store_bytes(win + r_dst + j, v, r_len - j);
                }
                serve_mask = ready_mask & ~wave::half_of(runs_w);
            }
        }
#endif
        // Small batches (round 4): a wavefront that decodes one page lends the idle upper half to the teams -- twice the lanes
        // per long piece.  All 64 lanes work in the one record of the wavefront (PageRecord<true>); the upper lanes take the
        // pieces' fields from the lower half's lanes.
        Team t;
        uint32_t t_pk, t_dist, t_src, t_dst, team_mask = serve_mask;
        uint8_t* t_lds = win;
        const uint8_t* t_stg = reinterpret_cast<const uint8_t*>(stage);
        if (solo) {
            const uint32_t lane = wave::lane_id();
            team_mask = wave::bcast(serve_mask, 0u);
            t = make_team64(team_mask, lane);
            t_pk = wave::bcast(packed, t.job); t_dist = wave::bcast(dist, t.job);
            t_src = wave::bcast(src_idx, t.job); t_dst = wave::bcast(dst_idx, t.job);
        } else {
            t = make_team(serve_mask, sl);
            t_pk = wave::half_shfl(packed, t.job); t_dist = wave::half_shfl(dist, t.job);
            t_src = wave::half_shfl(src_idx, t.job); t_dst = wave::half_shfl(dst_idx, t.job);
        }
        const uint32_t t_len = t_pk & 0x7FFu, t_far = (t_pk >> 11) & 0x7FFu;
        const uint8_t* t_stage = t_stg + ((t_pk >> 22) << 3);
        const uint8_t* t_win = t_lds + (int32_t)t_src;
        uint8_t* t_out = t_lds + t_dst;
        const bool act = t.serves && team_mask != 0u;
        const uint32_t t_pat = t_dist < t_len ? t_dist : t_len;
        const bool whole = t_far == 0u || t_far == t_pat;    // pattern in one place (window or staging area)
        const uint8_t* t_base = t_far ? t_stage : t_win;
        const bool overlap = t_dist < t_len;
        clk.lap(kPhLvShort);
#if BROTLIG_TUNE_POW2_OVERLAP
        // Every served piece a run with a period that divides 8 (a repeated byte, 16-bit sample, 32 / 64-bit word; pattern in one place): every
        // 8-byte chunk of it is the same word -- read once, stored by the team, no remainder per chunk (byte runs: config 2).
        const uint64_t act_w = wave::ballot64(act);
        const uint64_t p8_w = act_w & wave::ballot_lt(t_dist, t_len) & wave::ballot_lt_k<9u>(t_dist) & wave::ballot_eq0(t_dist & (t_dist - 1u)) &
                              (wave::ballot_eq0(t_far) | wave::ballot_eq(t_far, t_pat));
        if (act_w != 0ull && p8_w == act_w) {
            uint64_t v = 0;
            if (act) v = pattern_source8(t_base, t_dist, 0u);
            for (uint32_t c = t.member; wave::any(act && 8u * c < t_len); c += 1u << t.log2_size) {
                const uint32_t j = 8u * c;
                if (act && j < t_len) store_bytes(t_out + j, v, t_len - j);
            }
            wave::sync();
        } else
#endif
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
        todo_w &= ~ready_w;
        wave::sync();
    }
}

// ---- stage: per-page delta decode of the colour sub-streams (PageDecoder.cpp:446-471): a running byte sum over each
// colour range inside the page, in place in global memory, for the halves with `do_delta`.  16 bytes per lane and step,
// 512 contiguous bytes per half-wave: byte prefix inside the lane's chunk, half-wave scan of the chunk totals, running
// carry from step to step.
__device__ __forceinline__ void delta_decode_page(const PageJob& job, bool do_delta, uint32_t sl)
{
    if (!wave::any(do_delta)) return;
    wave::global_fence();                       // the page's own stores first
#if BROTLIG_TUNE_DELTA_UNROLL
#pragma unroll
#endif
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
#pragma unroll
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
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) w[k] = byte_add(w[k], add);
            if (full) {
                __builtin_memcpy(__builtin_assume_aligned(job.out + pos, 16), w, 16);
            } else {
#pragma unroll
                for (uint32_t i = 0; i < 16u; ++i)
                    if (pos + i >= lo && pos + i < hi) job.out[pos + i] = (uint8_t)(w[i >> 2] >> (8u * (i & 3u)));
            }
            carry = (carry + wave::half_shfl(incl, 31u)) & 0xFFu;
        }
    }
}

// ---- stage: sources of copies that lie below the output window ("far": in global memory, flushed by an earlier group).
// The first far_len bytes of a piece's pattern are far.  A piece that lies below the window as a whole, does not overlap
// itself and is at most kShortCopy bytes long (far_len == plen) never touches the staging area: its own lane fetches it
// and its bytes go from registers straight to their place in the window (`direct`).  Pieces of 8 bytes and more are
// covered by 8-byte chunks at offsets 0, 8, 16, 24 clipped to plen - 8 (the last chunk ends exactly at the piece's end
// and overlaps its predecessor); shorter ones by one load and a split store.  Everything else that reaches below the
// window is staged (8-byte aligned slots of the staging area, offsets by a half-wave scan): longer pieces, and patterns
// that straddle the window boundary.  Staged pieces of up to kShortCopy bytes are fetched by their own lane too; as soon
// as one is longer, all staged pieces get teams of lanes (two chunks per lane at once, the rest at store time).
// fetch_far_sources issues the loads and returns without waiting; store_far_sources puts the bytes where they belong.
struct FarSources {
    uint64_t fe0, fe1, fe2, fe3;    // own-lane chunks
    uint64_t te0, te1;              // team chunks
    Team     team;
    uint32_t t_src, t_len, t_stage; // the team's piece: page position of its source, far bytes, staging offset
    uint32_t stage_off;             // this lane's piece: 8-byte aligned offset into the staging area
    bool     direct, staged, any_staged, teams;
    uint64_t direct_w;              // `direct` of every lane (wave-wide mask)
};
__device__ __forceinline__ FarSources fetch_far_sources(const uint8_t* out, const uint8_t* win, uint32_t src_idx, bool near_direct,
                                                        uint32_t plen, uint32_t psrc, uint32_t far_len, uint32_t sl)
{
    FarSources f;
    // (the questions as lane masks: see wave::ballot_gt)
    const uint64_t far_w = wave::ballot_ne0(far_len);
    f.direct_w = far_w & wave::ballot_eq(far_len, plen) & wave::ballot_lt_k<kShortCopy + 1u>(plen);
    if (BROTLIG_TUNE_EARLY_NEAR) f.direct_w |= wave::ballot64(near_direct);
    const uint64_t staged_w = far_w & ~f.direct_w;
    f.direct = wave::from_mask(f.direct_w);
    f.staged = wave::from_mask(staged_w);
    const uint32_t stage_len = f.staged ? (far_len + 7u) & ~7u : 0u;
    f.any_staged = staged_w != 0ull;
    f.stage_off = 0;
    if (f.any_staged) f.stage_off = wave::half_scan_incl(stage_len) - stage_len;
    f.teams = f.any_staged && (staged_w & wave::ballot_gt_k<kShortCopy>(far_len)) != 0ull;
    f.fe0 = f.fe1 = f.fe2 = f.fe3 = f.te0 = f.te1 = 0;       // (leaving these unset saves six moves a group and costs 21 spilled values: measured, not done)
    f.team = Team{5u, 0u, 0u, false};
    f.t_src = f.t_len = f.t_stage = 0;
    const uint32_t clip8 = plen >= 8u ? plen - 8u : 0u;
    if (far_len != 0u && (f.direct || !f.teams)) {        // (conditional on purpose: unconditional clipped chunks cost 1.7 % -- more lane-loads on the memory path)
        const uint8_t* s8 = out + psrc;
        const uint32_t lim = f.direct ? clip8 : 24u;            // a staged piece keeps plain offsets
        f.fe0 = load_u64u(s8);
        if (far_len > 8u) f.fe1 = load_u64u(s8 + min_u32(8u, lim));
        if (far_len > 16u) f.fe2 = load_u64u(s8 + min_u32(16u, lim));
        if (far_len > 24u) f.fe3 = load_u64u(s8 + min_u32(24u, lim));
    }
    if (near_direct) {      // the same from the window: the source lies below the group, so it is final, and whole in LDS
        const uint8_t* s8 = win + src_idx;
        f.fe0 = load_u64u(s8);
        if (plen > 8u) f.fe1 = load_u64u(s8 + min_u32(8u, clip8));
        if (plen > 16u) { f.fe2 = load_u64u(s8 + min_u32(16u, clip8)); f.fe3 = load_u64u(s8 + clip8); }
    }
    if (f.teams) {
        const uint32_t staged_mask = wave::half_of(staged_w);
        f.team = make_team(staged_mask, sl);
        f.t_src = wave::half_shfl(psrc, f.team.job); f.t_len = wave::half_shfl(far_len, f.team.job);
        f.t_stage = wave::half_shfl(f.stage_off, f.team.job);
        f.team.serves = f.team.serves && staged_mask != 0u;
        const uint32_t tsz = 1u << f.team.log2_size;
        if (f.team.serves && 8u * f.team.member < f.t_len) f.te0 = load_u64u(out + f.t_src + 8u * f.team.member);
        if (f.team.serves && 8u * (f.team.member + tsz) < f.t_len) f.te1 = load_u64u(out + f.t_src + 8u * (f.team.member + tsz));
    }
    return f;
}
__device__ __forceinline__ void store_far_sources(uint8_t* win, uint64_t* stage, const uint8_t* out, const FarSources& f, uint32_t plen, uint32_t far_len, uint32_t dst_idx)
{
    const uint32_t clip8 = plen >= 8u ? plen - 8u : 0u;
    if (f.direct) {
        uint8_t* d = win + dst_idx;
        if (plen >= 8u) {                        // (conditional on purpose: many lanes take part, and a byte-misaligned LDS store costs a cycle per lane)
            __builtin_memcpy(d, &f.fe0, 8);
            if (plen > 8u) __builtin_memcpy(d + min_u32(8u, clip8), &f.fe1, 8);
            if (plen > 16u) __builtin_memcpy(d + min_u32(16u, clip8), &f.fe2, 8);
            if (plen > 24u) __builtin_memcpy(d + clip8, &f.fe3, 8);
        } else store_bytes(d, f.fe0, plen);
    }
    if (f.any_staged) {
        if (!f.teams) {
            if (f.staged) {
                uint64_t* st = &stage[f.stage_off >> 3];
                st[0] = f.fe0;
                if (far_len > 8u) st[1] = f.fe1;
                if (far_len > 16u) st[2] = f.fe2;
                if (far_len > 24u) st[3] = f.fe3;
            }
        } else {
            const uint32_t tsz = 1u << f.team.log2_size;
            if (f.team.serves && 8u * f.team.member < f.t_len) stage[(f.t_stage >> 3) + f.team.member] = f.te0;
            if (f.team.serves && 8u * (f.team.member + tsz) < f.t_len) stage[(f.t_stage >> 3) + f.team.member + tsz] = f.te1;
            for (uint32_t c = f.team.member + 2u * tsz; wave::any(f.team.serves && 8u * c < f.t_len); c += tsz) {
                if (f.team.serves && 8u * c < f.t_len) stage[(f.t_stage >> 3) + c] = load_u64u(out + f.t_src + 8u * c);
            }
        }
    }
}

// ---- stage: which earlier pieces of the group does my copy read?  The pieces of a group are consecutive commands (every
// command has at least one byte), each starting at byte `first_rel` of the group: a bitmap of the starts (bit p <=> a
// piece starts at group byte p) and the number of starts before each of its words answer "which piece owns byte x" with
// one popcount.  Returns the lanes (of the half) whose pieces own bytes of [psrc, src_end) inside the group and come
// before me; everything below the group (page position gpos) is final.
template <class Clock, class G = GeoPair>
__device__ __forceinline__ uint32_t piece_dependencies(uint32_t* start_bits, uint8_t* start_cum, bool on, uint64_t in_group_w, uint32_t first_rel, uint32_t gpos,
                                                        uint32_t psrc, uint32_t src_end, bool has_piece, uint32_t sl, Clock& clk)
{
    const uint32_t piece_mask = wave::half_of(in_group_w);              // (in_group_w: which lanes have a piece in the group)
    const bool in_group = wave::from_mask(in_group_w);
    if (on && sl < G::kRoundMax / 32u) start_bits[sl] = 0u;
    wave::sync();
    if (in_group) atomicOr(&start_bits[first_rel >> 5], 1u << (first_rel & 31u));
    wave::sync();
    {
        const bool rd = on && sl < G::kRoundMax / 32u;
        const uint32_t w = rd ? start_bits[sl] : 0u;
        const uint32_t cw = wave::half_scan_incl((uint32_t)__popc(w));
        if (rd) start_cum[sl] = (uint8_t)(cw - (uint32_t)__popc(w));
    }
    wave::sync();
    clk.lap(kPhBitmaps);
    uint32_t m = 0;
    if (has_piece && src_end > gpos) {
        const uint32_t first_piece = ctz_u32(piece_mask);
        const uint32_t hi_rel = src_end - 1u - gpos;
        const uint32_t hi = start_cum[hi_rel >> 5] + (uint32_t)__popc(start_bits[hi_rel >> 5] & (0xFFFFFFFFu >> (31u - (hi_rel & 31u))));
        uint32_t lo = 0;
        if (psrc > gpos) {
            const uint32_t lo_rel = psrc - gpos;
            lo = start_cum[lo_rel >> 5] + (uint32_t)__popc(start_bits[lo_rel >> 5] & (0xFFFFFFFFu >> (31u - (lo_rel & 31u)))) - 1u;
        }
        // ranks lo .. hi-1 among the group's pieces: rank r is lane first_piece + r; only pieces before me can be unfinished
        const uint32_t lo_l = first_piece + lo, hi_l = min_u32(first_piece + hi, sl);
        if (hi_l > lo_l) m = ((1u << hi_l) - 1u) & ~((1u << lo_l) - 1u);
    }
    return m;
}

}  // namespace brotlig
