"""Runs the HIP kernel SOURCE (brotli_g_sdk_amd/csrc/brotlig_kernels.h) on the CPU through the
fiber simulator in tests/sim and compares it with the oracle.  This is host-side logic coverage
for the GPU path: the same source is compiled by hipcc for gfx950 and checked on hardware by
tests/test_gpu_decode.py (-m gpu)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from brotli_g_sdk_amd import _build
from brotli_g_sdk_amd import datagen as D
from brotli_g_sdk_amd import encoder as E
from cases import plain_cases, precon_cases, raw_stress_cases, symbol_overflow_cases
from helpers import ROOT, oracle_decode

SIM_DIR = os.path.join(ROOT, "tests", "sim")
CSRC = os.path.join(ROOT, "brotli_g_sdk_amd", "csrc")


def build_sim(name, flags=()):
    # BROTLIG_SIM_FLAGS="-DBROTLIG_TUNE_X=1 ...": the whole simulator suite on a non-default build of the kernel source
    # (how the A/B variants of profiles/tools/ab_variants.sh are checked for bit-exactness before they go to the GPU box)
    extra = os.environ.get("BROTLIG_SIM_FLAGS", "").split()
    if extra:
        import hashlib
        name = name.replace(".so", "_" + hashlib.sha1(" ".join(extra).encode()).hexdigest()[:8] + ".so")
        flags = list(flags) + extra
    so = os.path.join(SIM_DIR, name)
    srcs = [os.path.join(SIM_DIR, f) for f in ("sim_decode.cpp", "sim_runtime.cpp")]
    deps = srcs + [os.path.join(SIM_DIR, f) for f in ("sim_runtime.h", "brotlig_wave_ops.h")] + \
        _build.kernel_headers()
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        tmp = f"{so}.{os.getpid()}.tmp"             # built aside and moved into place: pytest-xdist workers may get here together
        # (BROTLIG_SCHED_SPIN_LIMIT: the simulator runs the workgroups of the schedule kernel one after the other in ticket order -- a wait that
        # has to wait there is a broken ordering argument, and the kernel gives up after four looks instead of 2^25)
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-DBROTLIG_SCHED_SPIN_LIMIT=4", "-I", SIM_DIR, "-I", CSRC, "-o", tmp] + list(flags) + srcs)
        os.replace(tmp, so)
    L = ctypes.CDLL(so)
    L.sim_decode_batch.restype = ctypes.c_int
    L.sim_decode_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    L.sim_selftest.argtypes = [ctypes.c_void_p]
    # BROTLIG_SIM_DUO=1: every launch of the suite through the two-wavefronts-per-page kernel (tests/test_sim_duo.py runs its own set)
    L.sim_set_duo(1 if os.environ.get("BROTLIG_SIM_DUO", "0") == "1" else 0)
    return L


@pytest.fixture(scope="module")
def sim():
    return build_sim("libbrotlig_sim.so")


@pytest.fixture(scope="module")
def sim_small_caps():
    """The same kernel source with room for 24 ICP and 9 distance symbols in LDS: every page sends the rest of its
    symbols through the global-memory overflow."""
    return build_sim("libbrotlig_sim_smallcaps.so", ["-DBROTLIG_ICP_SYM_CAP=24", "-DBROTLIG_DIST_SYM_CAP=9"])


def run_batch(sim, streams, sizes, precon=False, grid=3):
    in_offs, pos = [], 0
    for s in streams:
        in_offs.append(pos)
        pos += (len(s) + 15) // 16 * 16
    buf = np.zeros(pos + 64, np.uint8)
    for o, s in zip(in_offs, streams):
        buf[o:o + len(s)] = s
    out_offs, opos = [], 0
    for n in sizes:
        out_offs.append(opos)
        opos += (n + 131071) // 131072 * 131072
    out = np.full(opos + 64, 0xCD, np.uint8)
    scratch = np.full(opos + 64, 0xEE, np.uint8) if precon else None
    io, oo = np.array(in_offs, np.uint64), np.array(out_offs, np.uint64)
    st = ctypes.c_uint32(0)
    sim.sim_decode_batch(buf.ctypes.data, pos, out.ctypes.data, opos, scratch.ctypes.data if precon else None,
                         io.ctypes.data, oo.ctypes.data, len(streams), grid, ctypes.byref(st))
    assert np.all(out[opos:] == 0xCD)
    return [out[o:o + n] for o, n in zip(out_offs, sizes)], st.value


@pytest.mark.parametrize("name,thunk,kw", plain_cases(), ids=[c[0] for c in plain_cases()])
def test_sim_plain(sim, name, thunk, kw):
    data = thunk()
    stream = E.encode(data, **kw)
    rc, ref = oracle_decode(stream)
    assert rc == 0
    outs, status = run_batch(sim, [stream], [len(data)])
    assert status == 0
    assert np.array_equal(outs[0], ref)


@pytest.mark.parametrize("name,thunk,pre", precon_cases(), ids=[c[0] for c in precon_cases()])
def test_sim_preconditioned(sim, name, thunk, pre):
    tex = thunk()
    stream = E.encode(tex, precondition=pre)
    rc, ref = oracle_decode(stream, out_size=len(tex))
    assert rc == 0
    outs, status = run_batch(sim, [stream], [len(tex)], precon=True)
    assert status == 0
    assert np.array_equal(outs[0], ref)


def test_sim_batch_of_streams(sim):
    """Several streams in one launch, odd page counts so halves pair pages of different streams."""
    datas = [D.text(65536 + 100, 1), D.runs(3 * 65536, 2), D.random_bytes(65536, 3), D.records(5 * 65536 + 1, 4), D.mixed(65536, 5)]
    streams = [E.encode(d) for d in datas]
    outs, status = run_batch(sim, streams, [len(d) for d in datas])
    assert status == 0
    for o, d in zip(outs, datas):
        assert np.array_equal(o, d)


def test_sim_one_page_per_wavefront_lends_the_upper_half_to_the_teams(sim):
    """Small batches: while a launch has no more pages than wavefronts every page decodes alone in its wavefront, and the idle
    upper half joins the copy teams of long pieces (round 4: 64 lanes per team set, `make_team64`).  Long runs, long record
    copies and long self-overlapping copies, with as many wavefronts as pages (all alone), with fewer (some paired, some alone)
    and with one wavefront for everything (all paired): the same bytes every time."""
    rng = np.random.default_rng(77)
    long_overlaps = np.concatenate([np.tile(rng.integers(0, 256, int(d), dtype=np.uint8), 1 + 900 // int(d))[:900] for d in rng.integers(1, 40, 80)])
    datas = [D.runs(3 * 65536, 21), D.records(2 * 65536 + 123, 22), long_overlaps, D.mixed(2 * 65536, 23)]
    streams = [E.encode(d) for d in datas]
    pages = sum((len(d) + 65535) // 65536 for d in datas)
    for grid in (pages, pages - 3, 1):
        outs, status = run_batch(sim, streams, [len(d) for d in datas], grid=grid)
        assert status == 0
        for o, d in zip(outs, datas):
            assert np.array_equal(o, d), grid


@pytest.mark.parametrize("name,thunk,kw", [c for c in raw_stress_cases() if c[0].endswith("64k")][-7:], ids=lambda v: v if isinstance(v, str) else "")
def test_sim_far_boundaries_one_page_per_wavefront(sim, name, thunk, kw):
    """The far / near boundary of the ONE-page layout (groups of 1 024 bytes, 1 040 bytes of history, a 5 120-byte window): sources just
    beyond the history, and sources that are in the window or not depending on when it last slid; as many wavefronts as pages."""
    data = thunk()[:3 * 65536 + 999]
    stream = E.encode(data, **kw)
    outs, status = run_batch(sim, [stream], [len(data)], grid=4)
    assert status == 0 and np.array_equal(outs[0], data)


def test_sim_bad_header_sets_status(sim):
    s = E.encode(D.text(70000, 1)); s[1] ^= 1
    outs, status = run_batch(sim, [s], [70000])
    assert status & 1


def test_sim_wave_primitives(sim):
    out = np.zeros(512, np.uint32)
    sim.sim_selftest(out.ctypes.data)
    v = out[320:384]
    for lane in range(64):
        base = lane & 32
        assert out[lane] == v[base:lane + 1].sum() == out[64 + lane]
        assert out[256 + lane] == v[base:base + 32].max()
        assert out[384 + lane] == v[base + (5 if base else 29)]
        assert out[448 + lane] == v[(base ^ 32):(base ^ 32) + 32].max()


def test_sim_pairing_policy(sim):
    """The schedule kernel compares the compressed sizes of neighbouring pages: a stream that mixes page
    kinds page by page lets the halves of a wavefront run free (threshold 1 quarter), a homogeneous one
    keeps them in step (4 quarters).  Either way the output is the same."""
    from brotli_g_sdk_amd import datagen as D
    sim.sim_last_policy.restype = ctypes.c_uint32
    sim.sim_set_order.argtypes = [ctypes.c_int]
    try:
        # (data, page schedule on?, expected threshold): in stream order the mixed pages differ from their
        # neighbours; once the schedule has grouped them by size they do not
        for data, order, want in ((D.mixed(65536 * 24, 5), 0, 1), (D.mixed(65536 * 24, 5), 1, 4), (D.text(65536 * 8, 6), 0, 4)):
            sim.sim_set_order(order)
            stream = E.encode(data)
            outs, status = run_batch(sim, [stream], [len(data)])
            assert status == 0 and np.array_equal(outs[0], data)
            assert sim.sim_last_policy() == want
    finally:
        sim.sim_set_order(1)


def test_sim_page_schedule_on_and_off(sim):
    """The page schedule (the schedule kernel's count and scatter phases: pages grouped into size buckets, dense first) only changes which
    half-wave decodes which page; the output is identical with and without it, for plain and
    pre-conditioned streams in one batch."""
    from brotli_g_sdk_amd import datagen as D
    datas = [D.mixed(65536 * 9 + 4321, 11), D.runs(65536 * 3, 12), D.random_bytes(65536 * 2 + 17, 13)]
    streams = [E.encode(d) for d in datas]
    tex = D.bc_texture(3, 64, 48, seed=14)
    streams.append(E.encode(tex, precondition=dict(format=3, width_blocks=64, height_blocks=48, swizzle=True, delta=True)))
    datas.append(tex)
    sim.sim_set_order.argtypes = [ctypes.c_int]
    try:
        for on in (0, 1):
            sim.sim_set_order(on)
            outs, status = run_batch(sim, streams, [len(d) for d in datas], precon=True)
            assert status == 0
            for o, d in zip(outs, datas):
                assert np.array_equal(o, d)
    finally:
        sim.sim_set_order(1)


@pytest.mark.parametrize("workers,fresh", [(1, 1), (2, 0), (5, 1), (3, 0)])
@pytest.mark.parametrize("schedule", ["proper", "page_order", "folded"])
def test_sim_schedule_kernel_by_tickets(sim, workers, fresh, schedule):
    """Round 6: the ONE kernel in front of the page decode (csrc/brotlig_schedule.h) with its workgroups as ticket holders -- prepare, count,
    scatter, policy items waiting for the phase before them -- instead of the one workgroup a batch this small gets from the host: plain,
    stored and pre-conditioned streams and a damaged one, with 1 .. 5 workgroups walking the pages, in a fresh workspace (garbage where the
    kernel keeps its counters: one workgroup initialises them, the others wait for its cookie) and in one the last batch left clean, under
    each of the three things the schedule can be.  Every record leads a page kernel to the right bytes; the damaged stream is named."""
    from fuzzcases import simple_code_one_symbol
    datas = [D.mixed(65536 * 9 + 4321, 11), D.runs(65536 * 3, 12), D.random_bytes(65536 * 2 + 17, 13), D.text(70000, 15)]
    streams = [E.encode(d) for d in datas]
    tex = D.bc_texture(3, 64, 48, seed=14)
    streams.append(E.encode(tex, precondition=dict(format=3, width_blocks=64, height_blocks=48, swizzle=True, delta=True)))
    datas.append(tex)
    bad, cap = simple_code_one_symbol()
    streams.append(bad)
    sizes = [len(d) for d in datas] + [cap]
    sim.sim_stream_status.restype = ctypes.c_uint32
    sim.sim_stream_status.argtypes = [ctypes.c_uint32]
    sim.sim_last_schedule_grid.restype = ctypes.c_uint32
    sim.sim_set_order_from_k.argtypes = [ctypes.c_uint32]
    # 21 pages: page order when the schedule proper starts at 1 024 pages; folded on 16 wavefronts (16 < 21 <= 32); the schedule proper on 4
    grid = {"proper": 4, "page_order": 4, "folded": 16}[schedule]
    sim.sim_set_order_from_k(1 if schedule == "page_order" else 0)
    sim.sim_set_schedule(workers, 1)
    sim.sim_set_fresh_workspace(fresh)
    try:
        for _ in range(2):                  # (the second batch finds what the first one left)
            outs, status = run_batch(sim, streams, sizes, precon=True, grid=grid)
            assert sim.sim_last_schedule_grid() == 1 + 2 * workers            # prepare, count and scatter items (the last of these runs the policy)
            assert status == 2
            assert [i for i in range(len(streams)) if sim.sim_stream_status(i)] == [len(streams) - 1]
            for o, d in zip(outs, datas):
                assert np.array_equal(o, d)
    finally:
        sim.sim_set_schedule(0, 0)
        sim.sim_set_fresh_workspace(0)
        sim.sim_set_order_from_k(0)


def test_sim_schedule_kernel_many_streams_by_tickets(sim):
    """More than 64 streams: the scan and finalize phases between prepare and count (chunks of 64 streams, their page totals summed by one
    item, added back by the next).  200 small streams, some of them stored, one refused by its header; 1, then 4 workgroups walk the pages."""
    rng = np.random.default_rng(7)
    makers = [D.text, D.records, D.samples16, D.runs, D.mixed, D.random_bytes]
    datas = [makers[i % 6](int(rng.integers(1, 140000)), 9000 + i) for i in range(200)]
    streams = [E.encode(d) for d in datas]
    hb = streams[77].copy(); hb[0] ^= 0x40; streams[77] = hb
    sim.sim_stream_status.restype = ctypes.c_uint32
    sim.sim_stream_status.argtypes = [ctypes.c_uint32]
    sim.sim_last_schedule_grid.restype = ctypes.c_uint32
    try:
        for workers in (1, 4):
            sim.sim_set_schedule(workers, 1)
            outs, status = run_batch(sim, streams, [len(d) for d in datas], grid=6)
            assert sim.sim_last_schedule_grid() == 1 + 1 + 1 + 2 * workers           # 4 chunks in one prepare item, scan, one finalize item, workers
            assert status == 1 and [i for i in range(200) if sim.sim_stream_status(i)] == [77]
            for i, (o, d) in enumerate(zip(outs, datas)):
                if i != 77:
                    assert np.array_equal(o, d), i
    finally:
        sim.sim_set_schedule(0, 0)


@pytest.mark.parametrize("name,thunk,kw", raw_stress_cases()[1::2], ids=[c[0] for c in raw_stress_cases()[1::2]])
def test_sim_far_boundary(sim, name, thunk, kw):
    """Copies from just beyond the on-chip history (far sources, straddling sources): logic check on the simulator;
    the memory-ordering side of it is tests/test_gpu_decode.py::test_far_copies_read_what_the_previous_group_flushed."""
    data = thunk()
    stream = E.encode(data, **kw)
    rc, ref = oracle_decode(stream)
    assert rc == 0 and np.array_equal(ref, data)
    outs, status = run_batch(sim, [stream], [len(data)])
    assert status == 0 and np.array_equal(outs[0], ref)


@pytest.mark.parametrize("name,thunk,kw", symbol_overflow_cases(), ids=[c[0] for c in symbol_overflow_cases()])
def test_sim_symbol_overflow(sim, name, thunk, kw):
    """More ICP / distance symbols in a page than the LDS arrays hold (product caps)."""
    data = thunk()
    stream = E.encode(data, **kw)
    rc, ref = oracle_decode(stream)
    assert rc == 0 and np.array_equal(ref, data)
    outs, status = run_batch(sim, [stream], [len(data)])
    assert status == 0 and np.array_equal(outs[0], ref)


def test_sim_small_caps_every_page_overflows(sim_small_caps):
    picks = [c for c in plain_cases() if c[0] in ("text", "mixed", "records_npostfix3", "skewed_long_codes", "mixed_128k_pages", "samples16")]
    streams, sizes, refs = [], [], []
    for name, thunk, kw in picks + symbol_overflow_cases()[:2]:
        data = thunk()
        streams.append(E.encode(data, **kw)); sizes.append(len(data)); refs.append(data)
    outs, status = run_batch(sim_small_caps, streams, sizes)
    assert status == 0
    for o, r in zip(outs, refs):
        assert np.array_equal(o, r)
    tex_name, tex_thunk, pre = precon_cases()[2]
    tex = tex_thunk()
    outs, status = run_batch(sim_small_caps, [E.encode(tex, precondition=pre)], [len(tex)], precon=True)
    assert status == 0 and np.array_equal(outs[0], tex)


def test_sim_simple_code_with_one_symbol_rejects_the_page(sim):
    """NSYM field 0 of a `simple` prefix code is undefined in the format (the reference indexes FixedCodelengths[-1],
    BrotligHuffmanTable.cpp:103): the kernel rejects the page -- status set, not one byte of it written -- like
    DecodeCPU (tests/test_cpu_decode.py), and a valid stream in the same launch is untouched by it."""
    from fuzzcases import simple_code_one_symbol
    bad, cap = simple_code_one_symbol()
    good_data = D.text(70000, 4)
    outs, status = run_batch(sim, [bad, E.encode(good_data)], [cap, len(good_data)])
    assert status & 2                                               # kStatusBadPage
    assert np.all(outs[0] == 0xCD)                                  # nothing of the rejected page reached the output
    assert np.array_equal(outs[1], good_data)


@pytest.mark.parametrize("duo,grid", [(0, 5), (0, 300), (1, 7)], ids=["two_pages_per_wavefront", "one_page_per_wavefront", "two_wavefronts_per_page"])
def test_sim_per_stream_status_names_the_damaged_streams(sim, duo, grid):
    """Round 5: the batch status is one OR-ed word for up to 4 096 streams; every stream now has a status word of its
    own (DcTable::status, read back by BrotligDecodeBatchStreamStatus).  64 streams, three damaged in three ways (refused when fetched,
    refused at page end, refused by the prepare kernel): exactly those three are named, with the right bit, in every kernel form, and the
    61 others are bit-exact."""
    from fuzzcases import damaged_batch_for_stream_status
    streams, sizes, datas, expect = damaged_batch_for_stream_status()
    sim.sim_stream_status.restype = ctypes.c_uint32
    sim.sim_stream_status.argtypes = [ctypes.c_uint32]
    sim.sim_set_duo(duo)
    try:
        outs, status = run_batch(sim, streams, sizes, grid=grid)
    finally:
        sim.sim_set_duo(1 if os.environ.get("BROTLIG_SIM_DUO", "0") == "1" else 0)
    assert status == 3
    got = {i: sim.sim_stream_status(i) for i in range(len(streams))}
    assert {i: v for i, v in got.items() if v} == expect
    for i, d in enumerate(datas):
        if d is not None:
            assert np.array_equal(outs[i], d), i


@pytest.mark.parametrize("n,bad", [(2003, 1234), (63, 5), (65, 64), (128, 127), (129, 128), (193, 0)])
def test_sim_batch_of_two_thousand_small_streams(sim, n, bad):
    """Thousands of small streams in one batch (an asset streamer's hand-over; the device test takes 20 011): the prepare kernel walks the
    headers 32 per step with a running page count, fetch_job searches the page-to-stream table, every stream has its status word.  2 003
    streams of 1 byte .. 5 KiB from 40 distinct ones, one of them with a damaged stream id: every other stream bit-exact and clean."""
    rng = np.random.default_rng(92)
    makers = [D.text, D.records, D.samples16, D.runs, D.mixed, D.random_bytes]
    base = [makers[i % 6](int(rng.integers(1, 5000)) if i % 4 else int(rng.integers(1, 30)), 4000 + i) for i in range(40)]
    enc = [E.encode(d) for d in base]
    pick = rng.integers(0, 40, n)                    # (the other sizes: around the 64 streams a workgroup of the prepare kernel takes)
    streams = [enc[k] for k in pick]
    hb = streams[bad].copy(); hb[1] ^= 0x01; streams[bad] = hb
    sim.sim_stream_status.restype = ctypes.c_uint32
    sim.sim_stream_status.argtypes = [ctypes.c_uint32]
    outs, status = run_batch(sim, streams, [len(base[k]) for k in pick], grid=9)
    assert status == 1
    assert [i for i in range(n) if sim.sim_stream_status(i)] == [bad] and sim.sim_stream_status(bad) == 1
    for i in range(n):
        if i != bad:
            assert np.array_equal(outs[i], base[pick[i]]), i


@pytest.mark.parametrize("grid", [1, 2, 5, 64, 1000])
def test_sim_decondition_cuts_the_batch_into_equal_runs(sim, grid):
    """The de-conditioning kernel takes the super-tiles of ALL pre-conditioned streams of a batch as one list (DcTable::super_base, a prefix
    the prepare kernels compute next to the page prefix) and gives every wavefront an equal run of it -- runs that begin and end anywhere:
    inside a texture, inside a mip, across plain streams that have no super-tiles, across several one-block textures.  Eleven streams (plain
    ones first, between and last; textures of one block, of many mips, wide, tall, of every format), with one wavefront for everything, with
    fewer wavefronts than super-tiles, and with far more wavefronts than super-tiles."""
    specs = [None, (3, 1, 1, 1), (1, 40, 24, 4), None, None, (5, 1, 1, 1), (2, 1, 1, 1), (4, 300, 6, 2), (3, 7, 90, 3), (1, 130, 130, 8), None]
    streams, sizes, want = [], [], []
    for k, sp in enumerate(specs):
        if sp is None:
            d = D.mixed(30000 + 999 * k, 600 + k)
            streams.append(E.encode(d)); sizes.append(len(d)); want.append(d)
        else:
            fmt, w, h, mips = sp
            tex = D.bc_texture(fmt, w, h, seed=700 + k, num_mips=mips)
            st = E.encode(tex, precondition=dict(format=fmt, width_blocks=w, height_blocks=h, num_mips=mips, swizzle=1, delta=1))
            rc, ref = oracle_decode(st, out_size=len(tex))
            assert rc == 0
            streams.append(st); sizes.append(len(tex)); want.append(ref)
    sim.sim_set_decond_grid(grid, 0)
    try:
        outs, status = run_batch(sim, streams, sizes, precon=True, grid=4)
    finally:
        sim.sim_set_decond_grid(3, 0)
    assert status == 0
    for k in range(len(specs)):
        assert np.array_equal(outs[k], want[k]), (k, specs[k])


@pytest.mark.parametrize("grid", [8, 300])
def test_sim_decondition_large_textures_go_to_gangs_of_256(sim, grid):
    """A batch whose textures have 1 024 super-tiles and more on average (4 MiB of BC3) is walked by gangs of 256 wavefronts that take a
    run's super-tiles in turn (large textures: HBM sees a few long streams); the same bytes with fewer wavefronts than a gang (one short
    gang) and with a full gang plus a short one.  Two textures, the rotation point of a run inside either."""
    specs = [(1, 1024, 256, 1), (4, 1024, 258, 2)]
    streams, sizes, want = [], [], []
    for k, (fmt, w, h, mips) in enumerate(specs):
        tex = D.bc_texture(fmt, w, h, seed=800 + k, num_mips=mips)
        st = E.encode(tex, precondition=dict(format=fmt, width_blocks=w, height_blocks=h, num_mips=mips, swizzle=1, delta=1))
        streams.append(st); sizes.append(len(tex)); want.append(tex)
    sim.sim_set_decond_grid(grid, 0)
    try:
        outs, status = run_batch(sim, streams, sizes, precon=True, grid=16)
    finally:
        sim.sim_set_decond_grid(3, 0)
    assert status == 0
    for k in range(len(specs)):
        assert np.array_equal(outs[k], want[k]), k


@pytest.mark.parametrize("grid,what", [(2, "page order"), (4, "folded"), (6, "folded"), (7, "one page per wavefront")])
def test_sim_schedule_modes_of_a_small_batch(sim, grid, what):
    """What the schedule kernel writes for a batch below the schedule's own threshold (the host: 12 288 pages; here raised above the batch
    through sim_set_order_from_k): page order -- unless the batch has more pages than the page kernel has wavefronts and at most twice as
    many, then the schedule FOLDED, front and back in turn, so that the two halves of a wavefront get the densest and the lightest page
    (schedule_mode in brotlig_kernels.h, late round 5).  Seven pages of very different cost (text, runs, stored, a short last page) on 2, 4,
    6 and 7 wavefronts: the same bytes."""
    datas = [D.text(65536, 31), D.runs(65536, 32), D.random_bytes(65536, 33), D.records(65536 + 9000, 34), D.samples16(65536, 35), np.zeros(700, np.uint8)]
    streams = [E.encode(d) for d in datas]
    sim.sim_set_order_from_k(60)
    try:
        outs, status = run_batch(sim, streams, [len(d) for d in datas], grid=grid)
    finally:
        sim.sim_set_order_from_k(0)
    assert status == 0
    for o, d in zip(outs, datas):
        assert np.array_equal(o, d), what


def test_sim_host_rule_exactly_one_kernel_decodes(sim):
    """The host launches BOTH page kernels when the output size leaves the page count open, and the device decides (DecodeArgs::duo_limit
    against the page count the prepare kernel found); it leaves the pairing policy at 0 when no two pages can meet (csrc/brotlig_hip.hip enqueue()).
    The same sequence on the simulator, with a small injected limit: page counts on both sides of it -- and AT it -- are decoded
    by exactly one of the two kernels (the page counter moves under one of them only), bit-exact either way."""
    for name in ("sim_counter_after_duo", "sim_counter_after_classic"):
        getattr(sim, name).restype = ctypes.c_uint32
    sim.sim_set_host_rule.argtypes = [ctypes.c_uint32]
    d = D.mixed(9 * 65536 + 100, 31)                                 # 10 pages
    stream = E.encode(d)
    try:
        for limit, grid, by_duo in ((10, 4, True), (9, 4, False), (11, 4, True), (10, 16, True), (3, 16, False)):
            sim.sim_set_host_rule(limit)
            outs, status = run_batch(sim, [stream], [len(d)], grid=grid)
            assert status == 0 and np.array_equal(outs[0], d), (limit, grid)
            duo_took, classic_took = sim.sim_counter_after_duo(), sim.sim_counter_after_classic()
            assert (duo_took >= 10 and classic_took == 0) if by_duo else (duo_took == 0 and classic_took >= 10), (limit, grid, duo_took, classic_took)
    finally:
        sim.sim_set_host_rule(0)
