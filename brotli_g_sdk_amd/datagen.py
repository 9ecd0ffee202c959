"""Seeded synthetic inputs for the BASELINE.json configs.

Everything is generated from seeds on whichever box runs the test/benchmark; nothing is shipped.
All generators return uint8 numpy arrays."""
import numpy as np

PAGE = 65536


def random_bytes(n, seed=0):
    """Config 1: uniform random bytes (encodes as stored pages)."""
    return np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8)


def runs(n, seed=1):
    """Config 2: zeros interleaved with byte runs (run length ~U[1,300], value ~U{0..3})."""
    rng = np.random.default_rng(seed)
    k = n // 100 + 16
    lens = rng.integers(1, 301, k)
    vals = rng.integers(0, 4, k).astype(np.uint8)
    vals[::2] = 0                                   # zeros interleaved with runs
    out = np.repeat(vals, lens)
    while len(out) < n:
        out = np.concatenate([out, out])
    return out[:n].copy()


_VOCAB_CACHE = {}


def _vocab(seed):
    """4096 pseudo-words as (flat bytes, offsets, lengths)."""
    if seed not in _VOCAB_CACHE:
        rng = np.random.default_rng(1000 + seed)
        letters = np.frombuffer(b"etaoinshrdlucmfwypvbgkqjxz", dtype=np.uint8)
        p = np.arange(1, 27, dtype=np.float64) ** -0.9
        p /= p.sum()
        lens = rng.integers(2, 11, 4096)
        flat = rng.choice(letters, int(lens.sum()), p=p)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
        _VOCAB_CACHE[seed] = (flat, offs, lens)
    return _VOCAB_CACHE[seed]


def text(n, seed=0):
    """Word soup: Zipf over a 4k-word vocabulary with punctuation and line breaks."""
    rng = np.random.default_rng(seed)
    flat, offs, lens = _vocab(seed % 4)
    nw = n // 3 + 64
    ranks = np.minimum(rng.zipf(1.25, nw) - 1, 4095)
    seps = rng.choice(np.frombuffer(b"     ,.\n;", dtype=np.uint8), nw)
    wl = lens[ranks] + 1                                  # word + separator
    starts = np.concatenate([[0], np.cumsum(wl)[:-1]])
    total = int(wl.sum())
    word_id = np.repeat(np.arange(nw), wl)
    within = np.arange(total) - starts[word_id]
    is_sep = within == (wl[word_id] - 1)
    src = offs[ranks][word_id] + np.minimum(within, lens[ranks][word_id] - 1)
    out = flat[src]
    out[is_sep] = seps[word_id[is_sep]]
    return out[:n].copy()


def records(n, seed=0):
    """Structured binary records: a repeating 16-64 byte template with mutated fields."""
    rng = np.random.default_rng(seed)
    out = np.empty(n + 64, dtype=np.uint8)
    pos = 0
    while pos < n:
        rec = int(rng.integers(16, 65))
        count = int(rng.integers(64, 1024))
        tmpl = rng.integers(0, 256, rec, dtype=np.uint8)
        block = np.tile(tmpl, count).reshape(count, rec)
        nf = int(rng.integers(1, 4))
        for _ in range(nf):                          # mutated fields: counters / small noise
            col = int(rng.integers(0, rec))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                block[:, col] = (np.arange(count) + int(rng.integers(0, 256))) & 0xFF
            elif kind == 1:
                block[:, col] = rng.integers(0, 8, count, dtype=np.uint8)
            else:
                block[:, col] = rng.integers(0, 256, count, dtype=np.uint8)
        flat = block.reshape(-1)
        take = min(len(flat), n + 64 - pos)
        out[pos:pos + take] = flat[:take]
        pos += take
    return out[:n].copy()


def samples16(n, seed=0):
    """Smooth 16-bit samples (random walk with small steps), little endian."""
    rng = np.random.default_rng(seed)
    m = n // 2 + 1
    steps = rng.integers(-6, 7, m)
    steps[rng.random(m) < 0.7] = 0
    walk = (np.cumsum(steps) + 20000).astype(np.int64) & 0xFFFF
    return walk.astype("<u2").view(np.uint8)[:n].copy()


def mixed_page(page_index, seed=0, n=PAGE):
    """One 'Silesia-like' page: the class is chosen per page with the BASELINE.md mix
    (40 % text, 25 % records, 20 % 16-bit samples, 10 % byte runs, 5 % random)."""
    rng = np.random.default_rng((seed << 20) ^ (page_index * 2654435761 & 0xFFFFFFFF))
    u = rng.random()
    s = int(rng.integers(0, 1 << 30))
    if u < 0.40:
        return text(n, s)
    if u < 0.65:
        return records(n, s)
    if u < 0.85:
        return samples16(n, s)
    if u < 0.95:
        return runs(n, s)
    return random_bytes(n, s)


def mixed(n, seed=0):
    pages = [mixed_page(i, seed, min(PAGE, n - i * PAGE)) for i in range((n + PAGE - 1) // PAGE)]
    return np.concatenate(pages)


def bc_texture(fmt, width_blocks, height_blocks, seed=0, num_mips=1, pitch_bytes=0, aligned=False):
    """Block-compressed texture bytes (BC1..BC5 layout): smooth endpoint gradients plus noisy
    index bits.  Returns (bytes, total_size) laid out mip after mip with the row pitch the
    reference derives (inc/common/BrotligDataConditioner.h:195-217)."""
    rng = np.random.default_rng(seed)
    block_bytes = {1: 8, 2: 16, 3: 16, 4: 8, 5: 16}[fmt]
    chunks = []
    w, h = width_blocks, height_blocks
    mw, mh = (w * 4) // 2, (h * 4) // 2
    for mip in range(num_mips):
        if mip > 0:
            w, h = (mw + 3) // 4, (mh + 3) // 4
            mw //= 2
            mh //= 2
        pitch = w * block_bytes
        if mip == 0 and pitch_bytes:
            pitch = pitch_bytes
        elif aligned:
            pitch = (pitch + 255) // 256 * 256
        tex = np.zeros((h, pitch), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        blocks = np.zeros((h, w, block_bytes), dtype=np.uint8)
        g = ((xx * 3 + yy * 5) // 4 + int(rng.integers(0, 64)))
        if fmt in (1, 2, 3):
            col_off = block_bytes - 8
            c0 = (g & 0x1F) | (((g // 2) & 0x3F) << 5) | (((g // 3) & 0x1F) << 11)
            c1 = ((g + 1) & 0x1F) | ((((g // 2) + 1) & 0x3F) << 5) | ((((g // 3)) & 0x1F) << 11)
            blocks[..., col_off + 0] = c0 & 0xFF
            blocks[..., col_off + 1] = (c0 >> 8) & 0xFF
            blocks[..., col_off + 2] = c1 & 0xFF
            blocks[..., col_off + 3] = (c1 >> 8) & 0xFF
            blocks[..., col_off + 4:col_off + 8] = rng.integers(0, 256, (h, w, 4), dtype=np.uint8) & \
                rng.choice(np.array([0x00, 0x55, 0xFF, 0x0F], dtype=np.uint8), (h, w, 1))
            if fmt == 2:
                blocks[..., 0:8] = rng.integers(0, 16, (h, w, 8), dtype=np.uint8) * 17
            if fmt == 3:
                blocks[..., 0] = (g + 40) & 0xFF
                blocks[..., 1] = (g + 8) & 0xFF
                blocks[..., 2:8] = rng.integers(0, 256, (h, w, 6), dtype=np.uint8) & 0x3F
        else:
            for base in range(0, block_bytes, 8):
                blocks[..., base + 0] = (g + 30 + base) & 0xFF
                blocks[..., base + 1] = (g + base) & 0xFF
                blocks[..., base + 2:base + 8] = rng.integers(0, 256, (h, w, 6), dtype=np.uint8) & 0x77
        tex[:, :w * block_bytes] = blocks.reshape(h, w * block_bytes)
        chunks.append(tex.reshape(-1))
    data = np.concatenate(chunks)
    return data


def tile_stream(stream, repeat):
    """Builds a stream whose page list is the page list of `stream` repeated `repeat` times.
    Pages are independent (no cross-page references, one set of prefix codes per page), so any
    concatenation with a rebuilt page table is a valid stream.  Only for
    non-preconditioned streams whose last page is full."""
    s = np.ascontiguousarray(stream, dtype=np.uint8)
    n = int(s[2]) | (int(s[3]) << 8)
    w1 = int(s[4:8].view("<u4")[0])
    assert (w1 >> 20) & 1 == 0 and ((w1 >> 2) & 0x3FFFF) == 0, "tile_stream needs full, unconditioned pages"
    table = s[8:8 + 4 * n].view("<u4").astype(np.int64)
    data = s[8 + 4 * n:]
    offs = table.copy()
    offs[0] = 0
    last_size = int(table[0])
    body_len = int(offs[-1]) + last_size if n > 1 else last_size
    body = data[:body_len]
    N = n * repeat
    assert N <= 65535
    new_offs = (offs[None, :] + (np.arange(repeat, dtype=np.int64) * body_len)[:, None]).reshape(-1)
    new_table = new_offs.astype("<u4")
    new_table[0] = last_size
    hdr = s[:8].copy()
    hdr[2] = N & 0xFF
    hdr[3] = N >> 8
    return np.concatenate([hdr, new_table.view(np.uint8), np.tile(body, repeat)])


# ---- real bytes (round 6): files that are already in the image ------------------------------------------------------
# The reference's own sample decodes USER FILES (sample/brotlig_cli.cpp:424-446); every other workload here is synthetic.  `files(n, seed)`
# returns n bytes read from files the ROCm image ships -- shared objects (x86 code and embedded GPU code objects), Python sources, C++
# headers, whatever /usr/share holds (text, locale catalogues, fonts' metadata) -- in a fixed, sorted order, a few hundred KiB per file, so
# that a 16 MiB sample already holds dozens of files of every kind.  The same image on the build container and on the GPU box gives the
# same bytes; `files_manifest` says what they were.  Nothing is copied into the repository.
FILE_ROOTS = (("elf", "/opt/rocm/lib", (".so",)), ("elf", "/usr/lib/x86_64-linux-gnu", (".so",)),
              ("python", "/usr/lib/python3.10", (".py",)), ("headers", "/opt/rocm/include", (".h", ".hpp")),
              ("share", "/usr/share", ()))
_FILE_LISTS = {}


def _file_list(kind_root):
    """Sorted regular files under a root (a name filter by suffix or, for shared objects, by '.so' anywhere in the name), largest first capped:
    at most 2 000 per root, by path order."""
    import os
    kind, root, suffixes = kind_root
    if kind_root not in _FILE_LISTS:
        found = []
        for dirpath, dirnames, names in os.walk(root, followlinks=False):
            dirnames.sort()
            for name in sorted(names):
                path = os.path.join(dirpath, name)
                if suffixes and not (name.endswith(suffixes) or (kind == "elf" and ".so." in name)):
                    continue
                try:
                    if os.path.islink(path) or not os.path.isfile(path) or os.path.getsize(path) < 4096:
                        continue
                except OSError:
                    continue
                found.append(path)
                if len(found) >= 2000:
                    break
            if len(found) >= 2000:
                break
        _FILE_LISTS[kind_root] = found
    return _FILE_LISTS[kind_root]


def files(n, seed=0, chunk=384 * 1024, manifest=None):
    """n bytes of real files: the next piece always from the root that has contributed the fewest bytes so far (shared objects are megabytes,
    sources kilobytes: by turns they would be 85 % of the sample), the next file of that root each time (the seed picks where in each list
    the walk starts), up to `chunk` bytes from the MIDDLE of a file (the head of a shared object is its symbol tables).  `manifest`, if a
    list, receives (kind, path, offset, bytes) per piece."""
    import os
    roots = [kr for kr in FILE_ROOTS if _file_list(kr)]
    if not roots:
        raise RuntimeError("datagen.files: none of the file roots exists on this machine")
    cursors = [(seed * 7919 + 13 * k) % len(_file_list(kr)) for k, kr in enumerate(roots)]
    out, have, guard = [], 0, 0
    given = [0] * len(roots)
    while have < n and guard < 100000:
        guard += 1
        k = min(range(len(roots)), key=lambda j: (given[j], (j + seed) % len(roots)))
        lst = _file_list(roots[k])
        path = lst[cursors[k] % len(lst)]; cursors[k] += 1
        try:
            size = os.path.getsize(path)
            take = min(chunk, size, n - have)
            off = ((size - take) // 2) & ~4095
            with open(path, "rb") as f:
                f.seek(off)
                b = np.frombuffer(f.read(take), dtype=np.uint8)
        except OSError:
            continue
        if len(b) == 0:
            continue
        out.append(b); have += len(b); given[k] += len(b)
        if manifest is not None:
            manifest.append((roots[k][0], path, int(off), int(len(b))))
    if have < n:
        raise RuntimeError("datagen.files: not enough readable bytes")
    return np.concatenate(out)[:n].copy()


def files_manifest_summary(manifest):
    """{kind: {"pieces", "bytes"}} plus a SHA-256 over the (path, offset, bytes) list: what a `files` sample was made of."""
    import hashlib
    by = {}
    h = hashlib.sha256()
    for kind, path, off, nb in manifest:
        d = by.setdefault(kind, {"pieces": 0, "bytes": 0})
        d["pieces"] += 1; d["bytes"] += nb
        h.update(f"{path}:{off}:{nb}\n".encode())
    return {"kinds": by, "pieces": len(manifest), "list_sha16": h.hexdigest()[:16]}
