/* cmd_stats.c -- diagnostics: what the decode kernel's round loop sees on a .brotlig stream.
 *
 * Builds the oracle (oracle/brotlig_oracle.c, test infrastructure) with its trace hook and replays every round through a
 * model of the fused kernel's bookkeeping (brotlig_kernels.h: 640-byte groups, 1344-byte window with 656 bytes of history,
 * far / direct / staged pieces, exact piece dependencies -> levels).  Prints one JSON object per stream.
 *   gcc -O2 -DBROTLIG_ORACLE_TRACE -o build/stats/cmd_stats profiles/tools/cmd_stats.c -lpthread
 *   build/stats/cmd_stats file.brotlig
 * Not part of the product or of the tests.
 */
#include "../../oracle/brotlig_oracle.c"
#include <stdio.h>

enum { kWin = 1344, kHist = 656, kRoundMax = 640, kShort = 32, kOwn = 128 };
static struct {
    uint64_t rounds, cmds, groups, lit_bytes, copy_bytes, lit_steps;
    uint64_t pieces_copy, far_direct, far_staged, near_nodep, near_dep, lit_pieces, lit_long;
    uint64_t levels, level_hist[40], ready_pieces, team_levels, overlap_pieces, short_pieces /* <8 */;
    uint64_t copy_len_hist[8], ins_hist[8], dist_hist[8];
    uint64_t multi_group_rounds, slides, ring_like;
    uint64_t lvl_bytes; /* sum over levels of max piece length in the level */
    uint64_t levels64; /* levels if 64 consecutive commands were one step */
    uint64_t simple_groups, simple_groups_long; /* groups whose copy pieces are all direct-eligible (<=32 bytes / any length) */
    uint64_t ingroup_pieces, long_pieces;
    uint64_t levels_fwd, fwd_pieces, p2_pieces;   /* with periodic forwarding (see below) */
    uint64_t levels_copyonly, false_dep_pieces;   /* round 5: dependencies counted only on the COPY bytes of earlier commands (their literal bytes are in the window before the levels start) */
} S;
static __thread uint32_t t_win_base, t_page_out;
/* round 5 ("two literals per LUT lookup"): how often do two consecutive literals of ONE sub-stream (j and j + 32 of a
 * round: what the kernel's literal loop decodes back to back) fit a table index of 8 / 10 / 12 bits together? */
static struct { uint64_t lits, pairs, fit8, fit10, fit12, len_sum, len_hist[16]; } LP;
/* how a symbol is decoded by the kernel's tables: one LUT read (code <= 8 bits), LUT + one read (longer, but every code under its 8-bit prefix has
 * the same length: "subtree"), or the canonical search (longer, lengths differ under the prefix) -- per table (ICP, distance, literal) */
static struct { uint64_t n[3], subtree[3], search[3]; uint64_t rounds_cmd, rounds_any_search[2]; } SY;
static __thread int t_round_search[2];
static void brotlig_oracle_trace_symbol(int k, uint32_t v, const uint16_t* codelens)
{
    uint32_t len = codelens[v];
    __sync_fetch_and_add(&SY.n[k], 1);
    if (len > 8) {
        uint32_t p = v & ~127u;
        if (codelens[p] == codelens[p | 127u]) __sync_fetch_and_add(&SY.subtree[k], 1);
        else { __sync_fetch_and_add(&SY.search[k], 1); if (k < 2) t_round_search[k] = 1; }
    }
}
static __thread uint32_t t_lane_len[32];
static void brotlig_oracle_trace_literal(uint32_t j, uint32_t len)
{
    __sync_fetch_and_add(&LP.lits, 1); __sync_fetch_and_add(&LP.len_sum, len); __sync_fetch_and_add(&LP.len_hist[len & 15], 1);
    if (j >= 32u && ((j >> 5) & 1u)) {          /* second literal of a pair of its lane: (0,32), (64,96), ... */
        uint32_t sum = t_lane_len[j & 31u] + len;
        __sync_fetch_and_add(&LP.pairs, 1);
        if (sum <= 8) __sync_fetch_and_add(&LP.fit8, 1);
        if (sum <= 10) __sync_fetch_and_add(&LP.fit10, 1);
        if (sum <= 12) __sync_fetch_and_add(&LP.fit12, 1);
    } else t_lane_len[j & 31u] = len;
}
static int bucket(uint32_t v) { return v == 0 ? 0 : v < 2 ? 1 : v < 4 ? 2 : v < 8 ? 3 : v < 16 ? 4 : v < 32 ? 5 : v < 128 ? 6 : 7; }
static int dbucket(uint32_t d) { return d < 8 ? 0 : d < 32 ? 1 : d < 128 ? 2 : d < 528 ? 3 : d < 2048 ? 4 : d < 8192 ? 5 : d < 32768 ? 6 : 7; }

static void brotlig_oracle_trace_round(const Cmd* q, uint32_t n, uint32_t out_pos, uint32_t litcount, uint32_t rlit)
{
    if (out_pos == 0) t_win_base = 0;
    __sync_fetch_and_add(&SY.rounds_cmd, 1);
    for (int k = 0; k < 2; ++k) { if (t_round_search[k]) __sync_fetch_and_add(&SY.rounds_any_search[k], 1); t_round_search[k] = 0; }
    uint32_t rel0[33], tot = 0;
    for (uint32_t k = 0; k < n; ++k) { rel0[k] = tot; tot += q[k].insert_len + q[k].copy_len; }
    rel0[n] = tot;
    __sync_fetch_and_add(&S.rounds, 1); __sync_fetch_and_add(&S.cmds, n);
    __sync_fetch_and_add(&S.lit_bytes, litcount);
    __sync_fetch_and_add(&S.lit_steps, n ? (rlit / n + 1) / 2 + 0 : 0);
    for (uint32_t k = 0; k < n; ++k) {
        __sync_fetch_and_add(&S.ins_hist[bucket(q[k].insert_len)], 1);
        if (q[k].copy_len) { __sync_fetch_and_add(&S.copy_len_hist[bucket(q[k].copy_len)], 1); __sync_fetch_and_add(&S.dist_hist[dbucket(q[k].dist)], 1);
                             __sync_fetch_and_add(&S.copy_bytes, q[k].copy_len); }
    }
    uint32_t ngroups = (tot + kRoundMax - 1) / kRoundMax;
    if (ngroups > 1) __sync_fetch_and_add(&S.multi_group_rounds, 1);
    for (uint32_t g = 0; g < ngroups; ++g) {
        uint32_t g0 = g * kRoundMax, g1 = tot < g0 + kRoundMax ? tot : g0 + kRoundMax;
        uint32_t gpos = out_pos + g0, gend = out_pos + g1;
        if (gend > t_win_base + kWin) { t_win_base = (gpos - kHist) & ~15u; __sync_fetch_and_add(&S.slides, 1); }
        __sync_fetch_and_add(&S.groups, 1);
        /* pieces */
        uint32_t lvl[32], plen[32], first = 32;
        uint32_t lvc[32];
        uint32_t lvf[32], p2[32], cdst[32];     /* model: a piece with a period of 1, 2 or 4 bytes whose pattern lies inside ONE earlier such piece takes its word from that piece's word (no wait) */
        int any = 0, not_simple = 0, not_simple_long = 0;
        for (uint32_t k = 0; k < n; ++k) {
            lvl[k] = 0; plen[k] = 0; lvf[k] = 0; p2[k] = 0; cdst[k] = 0; lvc[k] = 0;
            uint32_t r0 = rel0[k], t = q[k].insert_len + q[k].copy_len, cs = r0 + q[k].insert_len;
            if (!(r0 < g1 && r0 + t > g0)) continue;
            if (first == 32) first = k;
            uint32_t la = r0 > g0 ? r0 : g0, lb = cs < g1 ? cs : g1;
            if (lb > la) { __sync_fetch_and_add(&S.lit_pieces, 1); if (lb - la > kOwn) __sync_fetch_and_add(&S.lit_long, 1); }
            uint32_t ca = cs > g0 ? cs : g0, cb = r0 + t < g1 ? r0 + t : g1;
            if (!q[k].copy_len || cb <= ca) continue;
            uint32_t pl = cb - ca, pdst = out_pos + ca, psrc = pdst - q[k].dist;
            uint32_t pattern = pl < q[k].dist ? pl : q[k].dist, src_end = psrc + pattern;
            uint32_t far_len = psrc < t_win_base ? (pattern < t_win_base - psrc ? pattern : t_win_base - psrc) : 0;
            __sync_fetch_and_add(&S.pieces_copy, 1);
            if (pl < 8) __sync_fetch_and_add(&S.short_pieces, 1);
            {   /* direct-eligible: source final before the group starts, no self-overlap, in one place */
                int whole = far_len == 0 || far_len == pl;
                int below = src_end <= gpos && q[k].dist >= pl && whole;
                if (!below) { not_simple = 1; not_simple_long = 1; __sync_fetch_and_add(&S.ingroup_pieces, 1); }
                else if (pl > kShort) { not_simple = 1; __sync_fetch_and_add(&S.long_pieces, 1); }
            }
            if (far_len && far_len == pl && pl <= kShort) { __sync_fetch_and_add(&S.far_direct, 1); continue; }
            if (far_len) __sync_fetch_and_add(&S.far_staged, 1);
            plen[k] = pl;
            if (q[k].dist < pl && q[k].dist < 32) __sync_fetch_and_add(&S.overlap_pieces, 1);
            /* dependencies: earlier commands of the group owning bytes of [psrc, src_end) */
            uint32_t l = 1;
            if (src_end > gpos) {
                for (uint32_t j = first; j < k; ++j) {
                    uint32_t a = out_pos + (rel0[j] > g0 ? rel0[j] : g0), b = out_pos + (rel0[j + 1] < g1 ? rel0[j + 1] : g1);
                    if (a < src_end && b > psrc && lvl[j] + 1 > l) l = lvl[j] + 1;
                }
            }
            lvl[k] = l; any = 1;
            {   /* the same with the literal bytes of earlier commands taken as final */
                uint32_t lc = 1;
                if (src_end > gpos)
                    for (uint32_t j = first; j < k; ++j) {
                        if (!plen[j]) continue;
                        uint32_t cj = out_pos + (rel0[j] + q[j].insert_len > g0 ? rel0[j] + q[j].insert_len : g0), bj = out_pos + (rel0[j + 1] < g1 ? rel0[j + 1] : g1);
                        if (cj < src_end && bj > psrc && lvc[j] + 1 > lc) lc = lvc[j] + 1;
                    }
                lvc[k] = lc;
                if (lc < l) __sync_fetch_and_add(&S.false_dep_pieces, 1);
            }
            {   /* forwarding model */
                uint32_t d = q[k].dist, lf = 1; int fwd = 0;
                p2[k] = (d == 1 || d == 2 || d == 4) && d < pl && far_len == 0; cdst[k] = pdst;
                if (p2[k]) __sync_fetch_and_add(&S.p2_pieces, 1);
                if (src_end > gpos) {
                    for (uint32_t j = first; j < k; ++j) {
                        uint32_t a = out_pos + (rel0[j] > g0 ? rel0[j] : g0), b = out_pos + (rel0[j + 1] < g1 ? rel0[j + 1] : g1);
                        if (!(a < src_end && b > psrc)) continue;
                        /* pattern wholly inside the COPY bytes of j, both periodic: same level as j */
                        if (p2[k] && p2[j] && plen[j] && psrc >= cdst[j] && src_end <= cdst[j] + plen[j]) { if (lvf[j] > lf) lf = lvf[j]; fwd = 1; }
                        else if (lvf[j] + 1 > lf) lf = lvf[j] + 1;
                    }
                }
                lvf[k] = lf; if (fwd) __sync_fetch_and_add(&S.fwd_pieces, 1);
            }
            if (l == 1) __sync_fetch_and_add(far_len ? &S.ring_like : &S.near_nodep, 1); else __sync_fetch_and_add(&S.near_dep, 1);
        }
        uint32_t maxl = 0;
        for (uint32_t k = 0; k < n; ++k) if (lvl[k] > maxl) maxl = lvl[k];
        __sync_fetch_and_add(&S.levels, maxl);
        { uint32_t mc = 0; for (uint32_t k = 0; k < n; ++k) if (lvc[k] > mc) mc = lvc[k]; __sync_fetch_and_add(&S.levels_copyonly, mc); }
        { uint32_t mf = 0; for (uint32_t k = 0; k < n; ++k) if (lvf[k] > mf) mf = lvf[k]; __sync_fetch_and_add(&S.levels_fwd, mf); }
        __sync_fetch_and_add(&S.level_hist[maxl < 39 ? maxl : 39], 1);
        for (uint32_t l = 1; l <= maxl; ++l) {
            uint32_t mx = 0, cnt = 0;
            for (uint32_t k = 0; k < n; ++k) if (lvl[k] == l) { ++cnt; if (plen[k] > mx) mx = plen[k]; }
            __sync_fetch_and_add(&S.ready_pieces, cnt);
            __sync_fetch_and_add(&S.lvl_bytes, mx);
            if (mx > kOwn) __sync_fetch_and_add(&S.team_levels, 1);
        }
        (void)any;
        if (!not_simple) __sync_fetch_and_add(&S.simple_groups, 1);
        if (!not_simple_long) __sync_fetch_and_add(&S.simple_groups_long, 1);
    }
}

int main(int argc, char** argv)
{
    for (int i = 1; i < argc; ++i) {
        FILE* f = fopen(argv[i], "rb"); if (!f) { perror(argv[i]); return 1; }
        fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
        uint8_t* in = malloc(sz + 64); memset(in + sz, 0, 64);
        if (fread(in, 1, sz, f) != (size_t)sz) return 1;
        fclose(f);
        uint32_t osz = DecompressedSize(in);
        uint8_t* out = malloc((size_t)osz + 64);
        memset(&S, 0, sizeof S); memset(&LP, 0, sizeof LP); memset(&SY, 0, sizeof SY);
        int used = 0;
        int rc = brotlig_oracle_decode((uint32_t)sz, in, &osz, out, 1, &used);
        double pages = osz / 65536.0, R = (double)S.rounds, G = (double)S.groups;
        printf("{\"file\": \"%s\", \"rc\": %d, \"ratio\": %.3f, \"pages\": %.0f, \"rounds_per_page\": %.1f, \"cmds_per_round\": %.2f, \"bytes_per_cmd\": %.2f, "
               "\"lit_frac\": %.3f, \"lits_per_round\": %.1f, \"groups_per_round\": %.3f, \"slides_per_round\": %.3f, \"levels_per_group\": %.3f, "
               "\"copy_pieces_per_group\": %.2f, \"far_direct\": %.3f, \"far_staged\": %.3f, \"far_staged_l1\": %.3f, \"near_nodep\": %.3f, \"near_dep\": %.3f, "
               "\"simple_groups\": %.3f, \"simple_groups_anylen\": %.3f, \"ingroup_src_pieces_per_group\": %.2f, \"long_below_pieces_per_group\": %.2f, \"short_lt8\": %.3f, \"overlap_lt32\": %.3f, \"ready_per_level\": %.2f, \"team_level_frac\": %.3f, \"maxlen_per_level\": %.1f, \"lit_pieces_per_group\": %.2f, \"levels_per_group_with_periodic_forwarding\": %.3f, \"period_1_2_4_pieces\": %.3f, \"forwarded\": %.3f, \"levels_per_group_copy_bytes_only\": %.3f, \"pieces_with_a_literal_only_dependency\": %.3f,\n",
               argv[i], rc, (double)osz / sz, pages, R / pages, (double)S.cmds / R, (double)osz / S.cmds, (double)S.lit_bytes / osz, (double)S.lit_bytes / R,
               G / R, (double)S.slides / R, (double)S.levels / G, (double)S.pieces_copy / G,
               (double)S.far_direct / S.pieces_copy, (double)S.far_staged / S.pieces_copy, (double)S.ring_like / S.pieces_copy, (double)S.near_nodep / S.pieces_copy, (double)S.near_dep / S.pieces_copy,
               S.simple_groups / G, S.simple_groups_long / G, S.ingroup_pieces / G, S.long_pieces / G,
               (double)S.short_pieces / S.pieces_copy, (double)S.overlap_pieces / S.pieces_copy, (double)S.ready_pieces / (S.levels ? S.levels : 1),
               (double)S.team_levels / (S.levels ? S.levels : 1), (double)S.lvl_bytes / (S.levels ? S.levels : 1), (double)S.lit_pieces / G, (double)S.levels_fwd / G, (double)S.p2_pieces / S.pieces_copy, (double)S.fwd_pieces / S.pieces_copy, (double)S.levels_copyonly / G, (double)S.false_dep_pieces / S.pieces_copy);
        printf(" \"literal_pairs\": {\"literals\": %llu, \"mean_code_bits\": %.2f, \"pairs_fitting_8_bits\": %.3f, \"10_bits\": %.3f, \"12_bits\": %.3f},\n",
               (unsigned long long)LP.lits, LP.lits ? (double)LP.len_sum / LP.lits : 0.0, LP.pairs ? (double)LP.fit8 / LP.pairs : 0.0,
               LP.pairs ? (double)LP.fit10 / LP.pairs : 0.0, LP.pairs ? (double)LP.fit12 / LP.pairs : 0.0);
        printf(" \"symbol_paths\": {");
        { const char* nm[3] = {"icp", "dist", "lit"};
          for (int k = 0; k < 3; ++k) printf("\"%s\": {\"symbols\": %llu, \"one_more_read\": %.4f, \"canonical_search\": %.4f}, ", nm[k], (unsigned long long)SY.n[k],
                                             SY.n[k] ? (double)SY.subtree[k] / SY.n[k] : 0.0, SY.n[k] ? (double)SY.search[k] / SY.n[k] : 0.0); }
        printf("\"rounds_with_a_searching_lane_icp\": %.3f, \"dist\": %.3f},\n", SY.rounds_cmd ? (double)SY.rounds_any_search[0] / SY.rounds_cmd : 0.0,
               SY.rounds_cmd ? (double)SY.rounds_any_search[1] / SY.rounds_cmd : 0.0);
        printf(" \"level_hist\": [");
        for (int l = 0; l < 12; ++l) printf("%s%.3f", l ? ", " : "", S.level_hist[l] / G);
        printf("], \"ins_hist(0,1,2-3,4-7,8-15,16-31,32-127,128+)\": [");
        for (int l = 0; l < 8; ++l) printf("%s%.3f", l ? ", " : "", (double)S.ins_hist[l] / S.cmds);
        printf("], \"copy_hist\": [");
        for (int l = 0; l < 8; ++l) printf("%s%.3f", l ? ", " : "", (double)S.copy_len_hist[l] / S.cmds);
        printf("], \"dist_hist(<8,<32,<128,<528,<2k,<8k,<32k,more)\": [");
        for (int l = 0; l < 8; ++l) printf("%s%.3f", l ? ", " : "", (double)S.dist_hist[l] / S.cmds);
        printf("]}\n");
        free(in); free(out);
    }
    return 0;
}
