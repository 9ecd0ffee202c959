"""The split decode path (profiles/experiments/split_path/brotlig_split_kernels.h) on the CPU simulator: the entropy kernel's command /
literal arrays are assembled by a few lines of Python here and must give the encoder's input; the assembly kernel must give the
same bytes from the same arrays (test further down)."""
import ctypes

import numpy as np
import pytest

from brotli_g_sdk_amd import datagen as D
from brotli_g_sdk_amd import encoder as E
from cases import plain_cases, raw_stress_cases, symbol_overflow_cases
from test_sim_decode import build_sim

CMD_CAP = 40000
LIT_STRIDE = 131072 + 64


def _pages_up_to(cases, limit):
    """The experiment's slots hold pages of at most 128 KiB (the reference encoder's maximum); the product decodes the header's fourth page
    size too (cases *_256k_pages), the experiment was never asked to."""
    return [c for c in cases if c[2].get("page_size", 65536) <= limit]


PLAIN = _pages_up_to(plain_cases(), 131072)


@pytest.fixture(scope="module")
def sim():
    L = build_sim("libbrotlig_sim_split.so", split=True)
    L.sim_entropy_batch.restype = ctypes.c_int
    L.sim_entropy_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    return L


def layout(streams, sizes):
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
    return buf, pos, np.array(in_offs, np.uint64), np.array(out_offs, np.uint64), opos


def run_entropy(sim, streams, sizes):
    """Returns (per-page list of (cmds, lits, n, flags) in stream order, output buffer views, status)."""
    buf, in_bytes, io, oo, opos = layout(streams, sizes)
    out = np.full(opos + 64, 0xCD, np.uint8)
    page_sizes = [32768 << (int(s[4]) & 3) for s in streams]
    npages = [int(s[2]) | (int(s[3]) << 8) for s in streams]
    total = sum(npages)
    cmds = np.zeros((total, CMD_CAP + 1), np.uint64)
    lits = np.full((total, LIT_STRIDE), 0xEE, np.uint8)
    hdr = np.full((total, 2), 0xFFFFFFFF, np.uint32)
    st = ctypes.c_uint32(0)
    sim.sim_entropy_batch(buf.ctypes.data, in_bytes, out.ctypes.data, opos, io.ctypes.data, oo.ctypes.data, len(streams), 3,
                          cmds.ctypes.data, lits.ctypes.data, hdr.ctypes.data, CMD_CAP, LIT_STRIDE, ctypes.byref(st))
    return cmds, lits, hdr, out, [int(x) for x in oo], page_sizes, npages, st.value


def assemble(cmds, lits, n):
    """LZ77 assembly of one page from its packed commands (PageDecoder.cpp:209-233)."""
    m = (1 << 18) - 1
    c = [int(x) for x in cmds[:n + 1]]
    size = c[n] & m
    out = np.zeros(size, np.uint8)
    for k in range(n):
        o, lp, d = c[k] & m, (c[k] >> 18) & m, (c[k] >> 36) & m
        o2, lp2 = c[k + 1] & m, (c[k + 1] >> 18) & m
        ins, copy = lp2 - lp, (o2 - o) - (lp2 - lp)
        assert ins >= 0 and copy >= 0
        out[o:o + ins] = lits[lp:lp + ins]
        t = o + ins
        if copy:
            assert 0 < d <= t
            if d >= copy:
                out[t:t + copy] = out[t - d:t - d + copy]
            else:
                for j in range(copy):
                    out[t + j] = out[t - d + j]
    return out


def check_streams(sim, datas, streams):
    cmds, lits, hdr, out, oo, page_sizes, npages, status = run_entropy(sim, streams, [len(d) for d in datas])
    assert status == 0
    g = 0
    for s, (data, ps, npg) in enumerate(zip(datas, page_sizes, npages)):
        for i in range(npg):
            want = data[i * ps:(i + 1) * ps]
            n, flags = int(hdr[g, 0]), int(hdr[g, 1])
            if flags & 1:
                got = assemble(cmds[g], lits[g], n)
            else:                                                   # stored page: copied by the entropy kernel itself
                assert flags == 0
                got = out[oo[s] + i * ps:oo[s] + i * ps + len(want)]
            assert np.array_equal(got, want), (s, i)
            g += 1


@pytest.mark.parametrize("name,thunk,kw", PLAIN + raw_stress_cases()[:2] + symbol_overflow_cases(), ids=lambda v: v if isinstance(v, str) else "")
def test_entropy_kernel_arrays_assemble_to_the_source(sim, name, thunk, kw):
    data = np.ascontiguousarray(thunk(), dtype=np.uint8)
    if len(data) > 4 * 131072:
        data = data[:4 * 131072]
    check_streams(sim, [data], [E.encode(data, **kw)])


def test_entropy_kernel_batch_of_streams(sim):
    datas = [D.text(65536 + 100, 1), D.runs(3 * 65536, 2), D.random_bytes(65536, 3), D.records(2 * 65536 + 1, 4), D.mixed(65536, 5)]
    check_streams(sim, datas, [E.encode(d) for d in datas])


# ---- both kernels: entropy -> assembly (-> de-conditioning), against the encoder's input / the oracle -------------------------
def run_split(sim, streams, sizes, precon=False, mode=1):
    buf, in_bytes, io, oo, opos = layout(streams, sizes)
    out = np.full(opos + 64, 0xCD, np.uint8)
    scratch = np.full(opos + 64, 0xEE, np.uint8)
    npages = [int(s[2]) | (int(s[3]) << 8) for s in streams]
    total = sum(npages)
    cmds = np.zeros((total, CMD_CAP + 1), np.uint64)
    lits = np.full((total, LIT_STRIDE), 0xEE, np.uint8)
    hdr = np.full((total, 2), 0xFFFFFFFF, np.uint32)
    st = ctypes.c_uint32(0)
    sim.sim_set_scratch.argtypes = [ctypes.c_void_p]
    sim.sim_set_scratch(scratch.ctypes.data if precon else None)
    sim.sim_set_assemble(mode)
    try:
        sim.sim_entropy_batch(buf.ctypes.data, in_bytes, out.ctypes.data, opos, io.ctypes.data, oo.ctypes.data, len(streams), 3,
                              cmds.ctypes.data, lits.ctypes.data, hdr.ctypes.data, CMD_CAP, LIT_STRIDE, ctypes.byref(st))
    finally:
        sim.sim_set_assemble(0)
        sim.sim_set_scratch(None)
    assert np.all(out[opos:] == 0xCD)
    return [out[int(o):int(o) + n] for o, n in zip(oo, sizes)], st.value


@pytest.mark.parametrize("name,thunk,kw", PLAIN + raw_stress_cases() + symbol_overflow_cases(), ids=lambda v: v if isinstance(v, str) else "")
def test_split_path_plain(sim, name, thunk, kw):
    data = np.ascontiguousarray(thunk(), dtype=np.uint8)
    outs, status = run_split(sim, [E.encode(data, **kw)], [len(data)])
    assert status == 0 and np.array_equal(outs[0], data)


def test_split_path_preconditioned(sim):
    from cases import precon_cases
    from helpers import oracle_decode
    for name, thunk, pre in _pages_up_to(precon_cases(), 131072):
        tex = thunk()
        stream = E.encode(tex, precondition=pre)
        rc, ref = oracle_decode(stream, out_size=len(tex))
        assert rc == 0
        outs, status = run_split(sim, [stream], [len(tex)], precon=True)
        assert status == 0 and np.array_equal(outs[0], ref), name


def test_split_path_batch_and_random_streams(sim):
    from fuzzcases import random_plain
    datas = [D.text(65536 + 100, 1), D.runs(3 * 65536, 2), D.random_bytes(65536, 3), D.records(2 * 65536 + 1, 4), D.mixed(65536, 5)]
    outs, status = run_split(sim, [E.encode(d) for d in datas], [len(d) for d in datas])
    assert status == 0
    for o, d in zip(outs, datas):
        assert np.array_equal(o, d)
    for seed in range(60):
        data, kw = random_plain(seed)
        outs, status = run_split(sim, [E.encode(data, **kw)], [len(data)])
        assert status == 0 and np.array_equal(outs[0], data), seed


# ---- the same with the in-place assembly kernel (no LDS window; one wavefront per page) -------------------------------------------
@pytest.mark.parametrize("name,thunk,kw", PLAIN + raw_stress_cases() + symbol_overflow_cases(), ids=lambda v: v if isinstance(v, str) else "")
def test_split_path_global_assembly_plain(sim, name, thunk, kw):
    data = np.ascontiguousarray(thunk(), dtype=np.uint8)
    outs, status = run_split(sim, [E.encode(data, **kw)], [len(data)], mode=2)
    assert status == 0 and np.array_equal(outs[0], data)


def test_split_path_global_assembly_preconditioned_batch_random(sim):
    from cases import precon_cases
    from fuzzcases import random_plain
    from helpers import oracle_decode
    for name, thunk, pre in _pages_up_to(precon_cases(), 131072):
        tex = thunk()
        stream = E.encode(tex, precondition=pre)
        rc, ref = oracle_decode(stream, out_size=len(tex))
        assert rc == 0
        outs, status = run_split(sim, [stream], [len(tex)], precon=True, mode=2)
        assert status == 0 and np.array_equal(outs[0], ref), name
    datas = [D.text(65536 + 100, 1), D.runs(3 * 65536, 2), D.random_bytes(65536, 3), D.records(2 * 65536 + 1, 4), D.mixed(65536, 5)]
    outs, status = run_split(sim, [E.encode(d) for d in datas], [len(d) for d in datas], mode=2)
    assert status == 0
    for o, d in zip(outs, datas):
        assert np.array_equal(o, d)
    for seed in range(80):
        data, kw = random_plain(seed)
        outs, status = run_split(sim, [E.encode(data, **kw)], [len(data)], mode=2)
        assert status == 0 and np.array_equal(outs[0], data), seed


# ---- the page-in-LDS assembly kernel (one workgroup per page; pages of at most 64 KiB in this experiment) ---------------------
def _fits_lds(kw):
    return kw.get("page_size", 65536) <= 65536


@pytest.mark.parametrize("name,thunk,kw", [c for c in PLAIN + raw_stress_cases() + symbol_overflow_cases() if _fits_lds(c[2])],
                         ids=lambda v: v if isinstance(v, str) else "")
def test_split_path_page_in_lds_plain(sim, name, thunk, kw):
    data = np.ascontiguousarray(thunk(), dtype=np.uint8)
    outs, status = run_split(sim, [E.encode(data, **kw)], [len(data)], mode=3)
    assert status == 0 and np.array_equal(outs[0], data)


def test_split_path_page_in_lds_preconditioned_batch_random(sim):
    from cases import precon_cases
    from fuzzcases import random_plain
    from helpers import oracle_decode
    for name, thunk, pre in _pages_up_to(precon_cases(), 65536):
        tex = thunk()
        stream = E.encode(tex, precondition=pre)
        rc, ref = oracle_decode(stream, out_size=len(tex))
        assert rc == 0
        outs, status = run_split(sim, [stream], [len(tex)], precon=True, mode=3)
        assert status == 0 and np.array_equal(outs[0], ref), name
    datas = [D.text(65536 + 100, 1), D.runs(3 * 65536, 2), D.random_bytes(65536, 3), D.records(2 * 65536 + 1, 4), D.mixed(65536, 5)]
    outs, status = run_split(sim, [E.encode(d) for d in datas], [len(d) for d in datas], mode=3)
    assert status == 0
    for o, d in zip(outs, datas):
        assert np.array_equal(o, d)
    n = 0
    for seed in range(120):
        data, kw = random_plain(seed)
        if not _fits_lds(kw):
            continue
        outs, status = run_split(sim, [E.encode(data, **kw)], [len(data)], mode=3)
        assert status == 0 and np.array_equal(outs[0], data), seed
        n += 1
    assert n > 60
