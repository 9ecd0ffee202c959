"""Seeded random stream generators shared by the simulator tests (CPU) and the on-device differential and
negative tests (-m gpu): random data compositions x random encoder options x random pre-conditioning
parameters, and corrupted variants (bit flips, truncation, header damage)."""
import numpy as np

from brotli_g_sdk_amd import datagen as D
from brotli_g_sdk_amd import encoder as E


def compose(seed, n):
    """Random mix of literal stretches, byte runs, self-overlapping short periods, near and far repeats."""
    rng = np.random.default_rng(seed)
    out = np.empty(n + 70000, np.uint8)
    pos = 0
    while pos < n:
        kind = rng.integers(0, 8)
        if kind == 0 or pos < 16:                                   # fresh literals
            k = int(rng.integers(1, 200))
            out[pos:pos + k] = rng.integers(0, 256, k, dtype=np.uint8)
        elif kind == 1:                                             # byte run (distance 1)
            k = int(rng.integers(2, 3000))
            out[pos:pos + k] = out[pos - 1]
        elif kind == 2:                                             # short period, self-overlapping
            d = int(rng.integers(2, 40)); k = int(rng.integers(d, 2500))
            for i in range(k):
                out[pos + i] = out[pos + i - d]
        elif kind == 3:                                             # near repeat
            d = int(rng.integers(1, min(pos, 1500))) if pos > 1 else 1
            k = int(rng.integers(2, 64)); k = min(k, d)
            out[pos:pos + k] = out[pos - d:pos - d + k]
        elif kind == 4:                                             # far repeat, short
            d = int(rng.integers(1, pos + 1)); k = int(min(rng.integers(2, 40), d))
            out[pos:pos + k] = out[pos - d:pos - d + k]
        elif kind == 5:                                             # far repeat, long
            d = int(rng.integers(1, pos + 1)); k = int(min(rng.integers(40, 4000), d))
            out[pos:pos + k] = out[pos - d:pos - d + k]
        elif kind == 7:                                             # long literal stretch (crosses assembly groups)
            k = int(rng.integers(500, 5000))
            out[pos:pos + k] = rng.integers(0, 256, k, dtype=np.uint8)
        else:                                                       # skewed literals
            k = int(rng.integers(1, 400))
            out[pos:pos + k] = np.minimum(rng.geometric(0.3, k) - 1, 255)
        pos += k
    return out[:n].copy()


_FLAG_CHOICES = [0, 0, 0, E.NO_CODELEN_RLE, E.NO_RING_CODES, E.NO_LAZY, E.LITERALS_ONLY, E.FORCE_COMPLEX_TABLES,
                 E.SEARCH_DIST_PARAMS, E.OPTIMAL_PARSE, E.OPTIMAL_PARSE | E.SEARCH_DIST_PARAMS, E.NO_LAZY | E.NO_RING_CODES]


def random_plain(seed):
    """(data, encoder kwargs) for the seed: size 1 .. ~3 pages, any data class, any option set."""
    rng = np.random.default_rng(77000 + seed)
    page_size = int(rng.choice([32768, 65536, 65536, 131072]))
    n = int(rng.integers(1, 3 * page_size)) if rng.integers(0, 4) else int(rng.integers(1, 600))
    kind = int(rng.integers(0, 8))
    if kind == 0:
        data = compose(seed, n)
    elif kind == 1:
        data = D.text(n, seed)
    elif kind == 2:
        data = D.records(n, seed)
    elif kind == 3:
        data = D.samples16(n, seed)
    elif kind == 4:
        data = D.runs(n, seed)
    elif kind == 5:
        data = D.mixed(n, seed)
    elif kind == 6:
        data = np.minimum(rng.geometric(float(rng.uniform(0.05, 0.6)), n) - 1, 255).astype(np.uint8)
    else:
        data = D.random_bytes(n, seed) if rng.integers(0, 2) else np.full(n, int(rng.integers(0, 256)), np.uint8)
    npostfix = int(rng.integers(0, 4))
    kw = dict(page_size=page_size, npostfix=npostfix, ndirect_m=int(rng.integers(0, 16)),
              flags=int(_FLAG_CHOICES[int(rng.integers(0, len(_FLAG_CHOICES)))]))
    if rng.integers(0, 3) == 0:
        kw["max_chain"] = int(rng.integers(1, 64))
    return np.ascontiguousarray(data, dtype=np.uint8), kw


def random_precon(seed):
    """(texture bytes, precondition dict, encoder kwargs) with random format, size, mips, swizzle, delta, pitch."""
    rng = np.random.default_rng(88000 + seed)
    fmt = int(rng.integers(1, 6))
    bb = 8 if fmt in (1, 4) else 16
    w, h = int(rng.integers(1, 90)), int(rng.integers(1, 70))
    mips = int(rng.integers(1, 5)) if min(w, h) >= 8 else 1
    aligned = int(rng.integers(0, 2)) if mips > 1 else 0
    pitch = 0
    if mips == 1 and rng.integers(0, 3) == 0:
        pitch = w * bb + int(rng.integers(1, 40))
    pre = dict(format=fmt, width_blocks=w, height_blocks=h, num_mips=mips, swizzle=int(rng.integers(0, 2)),
               delta=int(rng.integers(0, 2)), pitch_d3d12_aligned=aligned, pitch_bytes=pitch)
    tex = D.bc_texture(fmt, w, h, seed=seed, num_mips=mips, aligned=bool(aligned), pitch_bytes=pitch)
    kw = dict(page_size=int(rng.choice([32768, 65536])), flags=int(rng.choice([0, 0, E.NO_LAZY, E.SEARCH_DIST_PARAMS])))
    return tex, pre, kw


def corrupt(stream, seed):
    """A damaged copy of `stream`: bit flips anywhere after the stream id, truncation, page-table damage or
    (for pre-conditioned streams) damage to the precondition header.  Returns (bytes, kind)."""
    rng = np.random.default_rng(99000 + seed)
    s = stream.copy()
    kind = ["flips", "flips", "truncate", "table", "precon_header", "page_header"][int(rng.integers(0, 6))]
    precon = bool((int(s[6]) >> 4) & 1)
    hdr = 16 if precon else 8
    npages = int(s[2]) | (int(s[3]) << 8)
    if kind == "precon_header" and not precon:
        kind = "flips"
    if kind == "flips":
        for _ in range(int(rng.integers(1, 8))):
            pos = int(rng.integers(2, len(s)))
            s[pos] ^= np.uint8(1 << int(rng.integers(0, 8)))
    elif kind == "truncate":
        s = s[:int(rng.integers(8, len(s)))].copy() if len(s) > 9 else s
    elif kind == "table":
        pos = hdr + int(rng.integers(0, 4 * npages))
        s[pos] ^= np.uint8(1 << int(rng.integers(0, 8)))
    elif kind == "precon_header":
        for _ in range(int(rng.integers(1, 4))):
            s[8 + int(rng.integers(0, 8))] ^= np.uint8(1 << int(rng.integers(0, 8)))
    else:                                                           # first bytes of a page: header + size table
        first = hdr + 4 * npages
        if first + 8 < len(s):
            s[first + int(rng.integers(0, 8))] ^= np.uint8(1 << int(rng.integers(0, 8)))
    return s, kind


def _first_page_substream0(stream):
    """Byte offset (in the stream) of sub-stream 0 of the first page of a plain single-page stream."""
    bw = lambda x: int(x).bit_length()
    npages = int(stream[2]) | (int(stream[3]) << 8)
    first = 8 + 4 * npages
    size = int.from_bytes(stream[8:12].tobytes(), "little") if npages == 1 else int.from_bytes(stream[12:16].tobytes(), "little")
    h = int.from_bytes(stream[first:first + 8].tobytes(), "little")
    base_bits, dsize_bits = bw((size + 31) // 32), bw(bw(size - 1))
    delta_bits = (h >> (8 + base_bits)) & ((1 << dsize_bits) - 1)
    return first + ((8 + base_bits + dsize_bits + 32 * delta_bits + 31) // 32) * 4


def simple_code_one_symbol():
    """(stream, output capacity): a valid stream whose first prefix-code description (ICP) is `simple`, with its
    NSYM field patched to 0 -- one symbol, which the format does not define (BrotligHuffmanTable.cpp:103 indexes
    FixedCodelengths[-1])."""
    for n in (40000, 65536, 20000, 3000):
        for data in (np.zeros(n, np.uint8), np.tile(np.arange(7, dtype=np.uint8), n // 7 + 1)[:n]):
            s = E.encode(data)
            at = _first_page_substream0(s)
            if int(s[at]) & 3 == 1:
                bad = s.copy()
                bad[at] &= np.uint8(0xF3)                           # NSYM - 1 field (bits 2..3) <- 0
                return bad, len(data)
    raise AssertionError("no input produced a simple ICP code")


def damaged_batch_for_stream_status(n=64, seed=5):
    """A batch of `n` small valid streams of every data class with THREE of them damaged in three different ways, for the per-stream
    status (BrotligDecodeBatchStreamStatus): returns (streams, output sizes, source bytes or None for the damaged ones,
    {index: expected kStatus bit}) -- 1 = header refused by the prepare kernel, 2 = a page failed.
      * a page table entry that points far outside the stream: the page is refused when it is fetched (fetch_job);
      * a prefix-code description the format does not define (simple_code_one_symbol): the page is refused after its table build, at
        page end -- the other flagging path;
      * a stream whose page table does not fit its own bytes (header says 40 000 pages): refused as a whole by the prepare kernel."""
    rng = np.random.default_rng(seed)
    makers = [D.text, D.records, D.samples16, D.runs, D.mixed, D.random_bytes]
    streams, sizes, datas = [], [], []
    for i in range(n):
        size = int(rng.integers(1, 3 * 65536))
        d = makers[i % len(makers)](size, 1000 + i)
        streams.append(E.encode(d)); sizes.append(len(d)); datas.append(d)
    expect = {}
    # (a) page table damage in a stream of at least two pages
    a = next(i for i in range(7, n) if sizes[i] > 65536 and i % len(makers) != 5)
    bad = streams[a].copy()
    bad[8 + 4:8 + 8] = np.frombuffer(np.uint32(0x7FFFFFF0).tobytes(), np.uint8)      # offset of page 1
    streams[a] = bad; datas[a] = None; expect[a] = 2
    # (b) an undefined code description in the first page
    b = a + 11
    sbad, cap = simple_code_one_symbol()
    streams[b] = sbad; sizes[b] = cap; datas[b] = None; expect[b] = 2
    # (c) a header whose page table cannot lie inside the stream
    c = a + 23
    hb = streams[c].copy()
    hb[2] = 40000 & 0xFF; hb[3] = 40000 >> 8
    streams[c] = hb; datas[c] = None; expect[c] = 1
    assert len({a, b, c}) == 3 and max(a, b, c) < n
    return streams, sizes, datas, expect
