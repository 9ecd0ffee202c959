"""Seeded decode cases shared by the oracle, simulator and GPU tests.  Each case is
(name, data-thunk, encoder kwargs); preconditioned cases carry a `precondition` dict."""
import numpy as np

from brotli_g_sdk_amd import datagen as D
from brotli_g_sdk_amd import encoder as E

N = 2 * 65536 + 4321          # two full pages and a short last page


def plain_cases():
    return [
        ("random_stored", lambda: D.random_bytes(65536, 0), {}),                      # BASELINE config 1
        ("zeros", lambda: np.zeros(70000, np.uint8), {}),
        ("one_byte", lambda: np.frombuffer(b"a", dtype=np.uint8), {}),
        ("short_text", lambda: np.frombuffer(b"abcabcabcabcabcabcabcabcabcabc" * 7, dtype=np.uint8), {}),
        ("runs", lambda: D.runs(N, 1), {}),
        ("text", lambda: D.text(N, 2), {}),
        ("records", lambda: D.records(N, 3), {}),
        ("samples16", lambda: D.samples16(N, 4), {}),
        ("mixed", lambda: D.mixed(6 * 65536, 5), {}),
        ("text_npostfix1", lambda: D.text(N, 2), dict(npostfix=1, ndirect_m=3)),
        ("records_npostfix2", lambda: D.records(N, 6), dict(npostfix=2, ndirect_m=7)),
        ("records_npostfix3", lambda: D.records(N, 7), dict(npostfix=3, ndirect_m=15)),
        ("text_no_rle", lambda: D.text(N, 8), dict(flags=E.NO_CODELEN_RLE)),
        ("records_no_ring", lambda: D.records(N, 9), dict(flags=E.NO_RING_CODES)),
        ("text_greedy", lambda: D.text(N, 10), dict(flags=E.NO_LAZY)),
        ("text_literals_only", lambda: D.text(N, 11), dict(flags=E.LITERALS_ONLY)),
        ("runs_complex_tables", lambda: D.runs(N, 12), dict(flags=E.FORCE_COMPLEX_TABLES)),
        ("all_stored", lambda: D.text(N, 13), dict(flags=E.FORCE_STORED)),
        ("mixed_optimal_parse", lambda: D.mixed(3 * 65536 + 500, 19), dict(flags=E.OPTIMAL_PARSE | E.SEARCH_DIST_PARAMS)),
        ("mixed_dist_param_search", lambda: D.mixed(5 * 65536, 18), dict(flags=E.SEARCH_DIST_PARAMS)),    # NPOSTFIX / NDIRECT vary per page
        # code-length tokens the reference's decoder accepts and its encoder never writes (16 straight after a 17-run: the last LITERAL length
        # survives the zeros; a zero run as a literal 0 and 16s: a 0 is a length like any other to 16 -- BrotligHuffmanTable.cpp:163-195)
        # (the same flag sets the header bits the reader skips, IS_DELTA on these plain pages -- dropped by the reader, PageDecoder.cpp:87-88 -- and lists
        # the symbols of simple codes in descending order: runs have trivial and simple codes)
        ("runs_decoder_corners", lambda: D.runs(N, 28), dict(flags=E.DECODER_CORNERS)),
        ("records_decoder_corners", lambda: D.records(N, 24), dict(flags=E.DECODER_CORNERS)),
        ("text_decoder_corners_complex", lambda: D.text(N, 25), dict(flags=E.DECODER_CORNERS | E.FORCE_COMPLEX_TABLES)),
        # eight clusters of eight equally likely byte values, eight unused values between them: equal lengths either side of every zero run
        ("clustered_alphabet_decoder_corners", lambda: (np.arange(64) // 8 * 16 + np.arange(64) % 8).astype(np.uint8)[
            np.random.default_rng(27).integers(0, 64, N)], dict(flags=E.DECODER_CORNERS)),
        ("records_smoothed_histograms", lambda: D.records(N, 22), dict(flags=E.SMOOTH_HISTOGRAMS)),       # code lengths with longer runs (more 16 / 17 tokens)
        ("text_32k_pages", lambda: D.text(N, 14), dict(page_size=32768)),
        ("mixed_128k_pages", lambda: D.mixed(3 * 65536 + 77, 15), dict(page_size=131072)),
        # the header's fourth page size (index 3 = 256 KiB: beyond the reference ENCODER's maximum, inc/common/BrotligConstants.h:85, but what its
        # decoders make of the index, inc/DataStream.h:73-75): positions of 18 bits, and distances only such a page can hold
        ("mixed_256k_pages", lambda: D.mixed(2 * 262144 + 777, 23), dict(page_size=262144)),
        ("far_matches_256k_pages", lambda: np.tile(D.random_bytes(150000, 31), 4)[:262144 + 99001], dict(page_size=262144)),
        ("skewed_long_codes", lambda: skewed(N, 16), {}),
        ("deep_literal_codes", lambda: np.minimum(np.random.default_rng(21).geometric(0.05, 200000) - 1, 255).astype(np.uint8), {}),   # 15-bit literal codes: their sub-tables overflow the page's second-level pool (canonical fallback)
        ("long_matches", lambda: np.tile(D.random_bytes(5000, 17), 40)[:N], {}),
        ("period_1_2_3", lambda: np.concatenate([np.full(30000, 7, np.uint8), np.tile(np.array([1, 2], np.uint8), 20000),
                                                 np.tile(np.array([9, 8, 7], np.uint8), 15000)]), {}),
        # short self-overlapping copies of every period 1..9 and every length 3..32, noise between them: the copy levels lay down periods of
        # 1, 2 and 4 bytes as one rotated 8-byte word (round 4), the others forward from their own output
        ("short_periodic_runs", short_periodic_runs, {}),
        ("short_periodic_runs_greedy", short_periodic_runs, dict(flags=E.NO_LAZY)),
        ("periodic_runs_to_40", lambda: short_periodic_runs(41), {}),       # with pieces above 32 bytes among them: those levels run in teams
        # ... and up to 56: the periods 1, 2 and 4 stay with their lane up to 48 bytes (kOverlapOwn, round 6: one word stored up to six times),
        # every other period and every longer run makes its level a team level
        ("periodic_runs_to_56", lambda: short_periodic_runs(57), {}),
        # long runs of every period next to long copies of earlier bytes, a literal or two between them: team levels that hold runs (periods 1, 2,
        # 4, 8: served first, by teams of their own -- round 6) AND other long pieces (periods 3, 5, 6, 7, 12, 20; plain copies from near and far)
        ("long_runs_beside_long_copies", long_runs_beside_long_copies, {}),
        ("long_runs_beside_long_copies_greedy", long_runs_beside_long_copies, dict(flags=E.NO_LAZY)),
    ]


def long_runs_beside_long_copies():
    rng = np.random.default_rng(606)
    base = rng.integers(0, 256, 3000, dtype=np.uint8)
    parts = [base]
    total = len(base)
    while total < 3 * 65536 + 1234:
        kind = int(rng.integers(0, 3))
        if kind < 2:                                                    # a run
            period = int(rng.choice([1, 2, 4, 8, 1, 2, 3, 5, 6, 7, 12, 20]))
            length = int(rng.integers(34, 420))
            pat = rng.integers(0, 256, period, dtype=np.uint8)
            piece = np.tile(pat, length // period + 2)[:period + length]
        else:                                                           # a copy of something earlier: near (inside the window) or far
            length = int(rng.integers(34, 320))
            whole = np.concatenate(parts) if len(parts) > 1 else parts[0]
            parts = [whole]
            back = int(rng.integers(length, min(len(whole), 600 if rng.random() < 0.7 else 40000)))
            piece = whole[len(whole) - back:len(whole) - back + length].copy()
        sep = rng.integers(0, 256, int(rng.integers(0, 3)), dtype=np.uint8)
        parts += [sep, piece]
        total += len(sep) + len(piece)
    return np.concatenate(parts)


def short_periodic_runs(length_end=33):
    rng = np.random.default_rng(404)
    parts = []
    for rep in range(6):
        for period in (1, 2, 4, 3, 8, 5, 2, 1, 4, 6, 7, 9):
            for length in range(3, length_end):
                pat = rng.integers(0, 256, period, dtype=np.uint8)
                parts.append(rng.integers(0, 256, int(rng.integers(1, 6)), dtype=np.uint8))       # noise: literals, and an odd alignment
                parts.append(np.tile(pat, (period + length) // period + 1)[:period + length])      # the pattern, then `length` bytes of its repeat
    return np.concatenate(parts)


def skewed(n, seed):
    """Geometric byte distribution: drives literal code lengths up to the 15-bit limit."""
    rng = np.random.default_rng(seed)
    return np.minimum(rng.geometric(0.35, n) - 1, 255).astype(np.uint8)


def precon_cases():
    out = []
    for name, fmt, w, h, mips, swz, delta, aligned, pitch, *page in [
        ("bc1_swz_delta", 1, 128, 128, 1, 1, 1, 0, 0),
        ("bc2_mips4", 2, 32, 32, 4, 1, 1, 0, 0),
        ("bc3_odd", 3, 64, 50, 1, 1, 1, 0, 0),
        ("bc4_noswz", 4, 256, 128, 1, 0, 1, 0, 0),
        ("bc5_odd_mips3", 5, 65, 33, 3, 1, 1, 0, 0),
        ("bc3_aligned_nodelta", 3, 64, 64, 2, 1, 0, 1, 0),
        ("bc3_mips6_aligned", 3, 37, 21, 6, 1, 1, 1, 0),
        ("bc5_pitch_pad", 5, 33, 17, 1, 1, 1, 0, 33 * 16 + 7),
        ("bc3_wide_swz_mips2", 3, 320, 6, 2, 1, 1, 0, 0),         # several 128-block steps per row pair, the last one partial (quad de-conditioning)
        ("bc1_mips_6_3_2", 1, 6, 6, 3, 1, 1, 0, 0),               # an even mip behind an odd one: its rows start 8 bytes off a 16-byte boundary
        ("bc4_swz_200x2", 4, 200, 2, 1, 1, 0, 0, 0),
        # round 5, the de-conditioning kernel's wide path (super-tiles of 2 x 128 blocks read as contiguous segments through LDS) next to its
        # gather path, for every block layout: row padding of whole blocks behind wide tiles, mips that stop being wide, a pitch that is
        # no multiple of the block size (gather only), eight-byte blocks
        ("bc2_wide_pitchpad", 2, 256, 4, 1, 1, 1, 0, 256 * 16 + 48),
        ("bc5_wide_mips3_aligned", 5, 384, 8, 3, 1, 1, 1, 0),
        ("bc3_wide_odd_pitch", 3, 128, 4, 1, 1, 0, 0, 128 * 16 + 7),
        ("bc4_wide_mips2", 4, 256, 6, 2, 1, 1, 0, 0),
        ("bc1_wide_tall", 1, 640, 34, 1, 1, 0, 0, 0),
        # every other page size under pre-conditioning (the delta ranges and the conditioned offsets of a page start at index x page size)
        ("bc1_pages_32k", 1, 300, 200, 1, 1, 1, 0, 0, 32768),
        ("bc3_mips4_pages_128k", 3, 200, 120, 4, 1, 1, 0, 0, 131072),
        ("bc5_aligned_mips3_pages_256k", 5, 129, 65, 3, 1, 1, 1, 0, 262144),
        ("bc4_noswz_pages_256k", 4, 256, 256, 2, 0, 1, 0, 0, 262144),
        # the precondition header's limits (inc/DataStream.h:89-98: width and height in blocks are 15 bits + 1): the widest and the tallest texture,
        # the widest with a full mip chain down to one block, and the smallest
        ("bc1_width_32768", 1, 32768, 3, 1, 1, 1, 0, 0),
        ("bc4_height_32768", 4, 3, 32768, 1, 1, 1, 0, 0),
        ("bc1_width_32768_mips16_aligned", 1, 32768, 1, 16, 1, 1, 1, 0),
        ("bc3_one_block", 3, 1, 1, 1, 1, 1, 0, 0),
        # narrow swizzled mips (a power of two below 128 columns, even height, no row padding): super-tiles of 128 / W row pairs, 256 consecutive
        # blocks of every sub-stream each -- whole groups through the wide path, the last group of a mip (fewer row pairs) through the gather, and
        # next to them the shapes that must NOT be taken for narrow: an odd height, padded rows, no swizzle, 128 columns exactly
        ("bc3_narrow_64x64", 3, 64, 64, 1, 1, 1, 0, 0),
        ("bc1_narrow_32x32_mips6", 1, 32, 32, 6, 1, 1, 0, 0),
        ("bc5_narrow_16x40", 5, 16, 40, 1, 1, 0, 0, 0),                 # 20 row pairs in groups of 8: two whole groups and four pairs
        ("bc4_narrow_2x200", 4, 2, 200, 1, 1, 1, 0, 0),                 # 100 row pairs in groups of 64
        ("bc2_narrow_64x10_mips3_aligned", 2, 64, 10, 3, 1, 1, 1, 0),   # 64 x 10 narrow (1 KiB rows), 32 x 5 odd, 16 x 2 with rows padded to 256 bytes
        ("bc3_narrow_64x6_pitchpad", 3, 64, 6, 1, 1, 1, 0, 64 * 16 + 16),
        ("bc1_narrow_64x64_noswizzle", 1, 64, 64, 1, 0, 1, 0, 0),
        ("bc5_128x8_exactly", 5, 128, 8, 2, 1, 1, 0, 0),                # 128 columns: the general wide path; its 64 x 4 mip: narrow, one whole group
        ("bc3_narrow_8x512_mips4", 3, 8, 512, 4, 1, 1, 0, 0),           # 8, 4, 2 columns narrow; the 1 x 64 mip is not
    ]:
        pre = dict(format=fmt, width_blocks=w, height_blocks=h, num_mips=mips, swizzle=swz, delta=delta,
                   pitch_d3d12_aligned=aligned, pitch_bytes=pitch)
        if page:
            pre["page_size"] = page[0]          # brotli_g_sdk_amd.encoder.encode takes it from here
        out.append((name, (lambda f=fmt, w=w, h=h, m=mips, a=aligned, p=pitch, s=len(out):
                           D.bc_texture(f, w, h, seed=100 + s, num_mips=m, aligned=bool(a), pitch_bytes=p)), pre))
    return out


def far_boundary(n, seed, dlo, dhi, lit=8, cpy=24):
    """Literal runs of `lit` fresh bytes followed by copies of `cpy` bytes from a distance in [dlo, dhi]: with the
    distances around the decoder's on-chip history (kHist = 656 bytes since round 4, 528 before) every copy reads bytes the window has just
    given up -- from global memory, where an earlier group of the same wavefront flushed them moments before --
    or straddles the window boundary."""
    rng = np.random.default_rng(seed)
    out = np.empty(n + lit + cpy + 64, np.uint8)
    out[:dhi + 1] = rng.integers(0, 256, dhi + 1, dtype=np.uint8)
    pos = dhi + 1
    while pos < n:
        out[pos:pos + lit] = rng.integers(0, 256, lit, dtype=np.uint8)
        pos += lit
        d = int(rng.integers(dlo, dhi + 1))
        out[pos:pos + cpy] = out[pos - d:pos - d + cpy]
        pos += cpy
    return out[:n].copy()


def raw_stress_cases():
    """Global read-after-write inside a wavefront: far copies whose sources were flushed by
    the previous assembly group.  (name, data thunk, encoder kwargs)."""
    out = []
    for k, (dlo, dhi, lit, cpy) in enumerate([(513, 560, 8, 24), (529, 544, 4, 28), (529, 1100, 8, 24), (520, 540, 1, 63),
                                               (528, 536, 16, 16), (1000, 1100, 8, 120),
                                               # round 4: groups of 640 bytes, 656 bytes of history
                                               (641, 700, 8, 24), (657, 672, 4, 28), (657, 1400, 8, 24), (648, 668, 1, 63),
                                               (656, 664, 16, 16),
                                               # round 4, one page per wavefront: groups of 1 024 bytes, 1 040 bytes of history, 5 120-byte window
                                               # (a source between 1 040 and ~4 100 bytes back is in the window or not depending on the last slide)
                                               (1025, 1100, 8, 24), (1041, 1056, 4, 28), (1030, 1050, 1, 63), (3900, 5300, 8, 56), (5100, 5200, 16, 120)]):
        for page in (65536, 131072):
            out.append((f"far_{dlo}_{dhi}_{lit}_{cpy}_{page >> 10}k",
                        (lambda a=dlo, b=dhi, l=lit, c=cpy, s=k: far_boundary(4 * 131072 + 777, 40 + s, a, b, l, c)),
                        dict(page_size=page)))
    return out


def many_command_shapes(n, seed):
    """Literal runs and copies whose lengths are spread log-uniformly over 0..200 and 2..600: about 300 distinct
    insert-and-copy symbols in a 128 KiB page (the benchmark's data classes use ~150 at most), more than the decoder
    keeps in LDS (kIcpSymCap = 255) -- the rest of the code's symbols are read from global memory."""
    rng = np.random.default_rng(seed)
    out = np.empty(n + 4096, np.uint8)
    out[:3000] = rng.integers(0, 256, 3000, dtype=np.uint8)
    pos = 3000
    while pos < n:
        lit = int(np.exp(rng.uniform(0, np.log(200)))) - 1
        out[pos:pos + lit] = rng.integers(0, 256, lit, dtype=np.uint8)
        pos += lit
        cpy = min(int(np.exp(rng.uniform(np.log(2), np.log(600)))), 600)
        d = int(rng.integers(cpy, min(pos, 60000))) if pos > cpy + 1 else pos
        out[pos:pos + cpy] = out[pos - d:pos - d + cpy]
        pos += cpy
    return out[:n].copy()


def many_distances(n, seed):
    """Short copies from distances spread over 10..400 and beyond: with NPOSTFIX 3 / NDIRECT 120 a page uses ~300
    distinct distance symbols, three times what the decoder keeps in LDS (kDistSymCap = 96)."""
    rng = np.random.default_rng(seed)
    out = np.empty(n + 64, np.uint8)
    out[:3000] = rng.integers(0, 256, 3000, dtype=np.uint8)
    pos = 3000
    while pos < n:
        lit = int(rng.integers(1, 6))
        out[pos:pos + lit] = rng.integers(0, 256, lit, dtype=np.uint8)
        pos += lit
        cpy = int(rng.integers(4, 10))
        d = int(rng.integers(10, 400)) if rng.integers(0, 4) else int(np.exp(rng.uniform(np.log(400), np.log(60000))))
        d = min(d, pos)
        out[pos:pos + cpy] = out[pos - d:pos - d + cpy]
        pos += cpy
    return out[:n].copy()


def symbol_overflow_cases():
    """Pages whose ICP / distance codes have more symbols than the decoder's LDS arrays hold."""
    return [
        ("many_command_shapes_128k", lambda: many_command_shapes(3 * 131072 + 100, 5), dict(page_size=131072)),
        ("many_distances_np3", lambda: many_distances(3 * 65536 + 100, 6), dict(npostfix=3, ndirect_m=15)),
        ("many_distances_np2_32k", lambda: many_distances(3 * 32768 + 100, 7), dict(npostfix=2, ndirect_m=15, page_size=32768)),
    ]
