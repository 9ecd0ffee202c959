"""Seeded fuzz of the kernel source on the CPU simulator: random compositions of literal runs,
short/long repeats at near and far distances, byte runs and self-overlapping copies, so that rounds
alternate between the windowed path and the global-memory path inside one page."""
import numpy as np
import pytest

from brotli_g_sdk_amd import encoder as E
from helpers import oracle_decode
from fuzzcases import compose, corrupt, random_plain, random_precon
from test_sim_decode import run_batch, sim  # noqa: F401  (fixture)


@pytest.mark.parametrize("seed", range(24))
def test_sim_fuzz(sim, seed):
    n = int(np.random.default_rng(1000 + seed).integers(1, 3 * 65536))
    data = compose(seed, n)
    kw = [dict(), dict(npostfix=1, ndirect_m=2), dict(flags=E.NO_LAZY), dict(page_size=32768), dict(flags=E.NO_RING_CODES)][seed % 5]
    stream = E.encode(data, **kw)
    rc, ref = oracle_decode(stream)
    assert rc == 0 and np.array_equal(ref, data)
    outs, status = run_batch(sim, [stream], [len(data)])
    assert status == 0
    assert np.array_equal(outs[0], ref)


@pytest.mark.parametrize("seed", range(12))
def test_sim_corrupt_streams_terminate(sim, seed):
    """The reference has no validation (undefined behaviour on corrupt pages); this
    decoder must at least terminate and stay inside its buffers.  Bit-flipped streams go through the
    kernel source on the simulator: any outcome is acceptable except a crash, a hang or a write past
    the output (checked by run_batch's guard bytes)."""
    from brotli_g_sdk_amd import datagen as D
    rng = np.random.default_rng(5000 + seed)
    data = [D.text, D.records, D.samples16, D.runs][seed % 4](65536 + 9000, seed)
    stream = E.encode(data).copy()
    for _ in range(int(rng.integers(1, 6))):
        pos = int(rng.integers(2, len(stream)))
        stream[pos] ^= np.uint8(1 << int(rng.integers(0, 8)))
    outs, status = run_batch(sim, [stream], [len(data)])
    assert len(outs[0]) == len(data)


@pytest.mark.parametrize("seed", range(0, 220, 11))
def test_sim_random_options(sim, seed):
    """A sample of the on-device differential cases (tests/test_gpu_differential.py) through the simulator."""
    data, kw = random_plain(seed)
    stream = E.encode(data, **kw)
    rc, ref = oracle_decode(stream)
    assert rc == 0 and np.array_equal(ref, data)
    outs, status = run_batch(sim, [stream], [len(data)])
    assert status == 0 and np.array_equal(outs[0], ref)


@pytest.mark.parametrize("seed", range(0, 60, 5))
def test_sim_random_precondition(sim, seed):
    tex, pre, kw = random_precon(seed)
    stream = E.encode(tex, precondition=pre, **kw)
    rc, ref = oracle_decode(stream, out_size=len(tex))
    assert rc == 0 and np.array_equal(ref, tex)
    outs, status = run_batch(sim, [stream], [len(tex)], precon=True)
    assert status == 0 and np.array_equal(outs[0], ref)


@pytest.mark.parametrize("seed", range(40))
def test_sim_corrupt_variants_stay_in_bounds(sim, seed):
    """Every corruption kind of fuzzcases.corrupt -- including damaged precondition headers, whose geometry
    fields drive the de-conditioning kernel's addressing -- terminates and writes nothing outside the output
    (run_batch checks the guard bytes behind it)."""
    if seed % 2:
        tex, pre, kw = random_precon(seed)
        stream, n, precon = E.encode(tex, precondition=pre, **kw), len(tex), True
    else:
        data, kw = random_plain(seed)
        stream, n, precon = E.encode(data, **kw), len(data), False
    bad, kind = corrupt(stream, seed)
    if bad[0] != 5 or bad[1] != 250:
        return
    outs, status = run_batch(sim, [bad], [n], precon=precon)
    assert len(outs[0]) == n


def test_sim_damaged_header_cannot_reach_a_neighbouring_stream(sim):
    """LastPageSize with bit 17 set claims a 128 KiB + 505 byte "short" page inside a 64 KiB page: the stream is
    rejected by the prepare kernel and the stream laid out behind it decodes bit-exactly."""
    d = np.frombuffer(bytes(range(256)) * 2, dtype=np.uint8)[:505].copy()
    bad = E.encode(d).copy()
    bad[6] |= 0x08
    good, kw = random_plain(3)
    outs, status = run_batch(sim, [bad, E.encode(good, **kw)], [len(d), len(good)])
    assert status != 0
    assert np.array_equal(outs[1], good)


@pytest.mark.parametrize("duo", [0, 1], ids=["one_wavefront", "two_wavefronts"])
@pytest.mark.parametrize("base", range(150000, 150320, 40))
def test_sim_damaged_batches_sized_by_their_own_headers(sim, base, duo):
    """What the device soak does (profiles/tools/soak.py): forty damaged streams and a valid one in ONE launch, every output region sized by
    what the damaged header claims.  Round 4's soak found two memory access faults this way that single streams with the true sizes never
    showed (an empty page outside its stream; a wrapped distance in the two-wavefront kernel).  Both kernels; any status, no fault (the
    simulator turns an out-of-bounds access into a crash), the valid neighbour bit-exact."""
    from brotli_g_sdk_amd.api import DecompressedSize
    streams, sizes = [], []
    for seed in range(base, base + 40):
        d, kw = random_plain(seed)
        bad, kind = corrupt(E.encode(d, **kw), seed)
        try:
            n = int(DecompressedSize(bad))
        except Exception:
            n = len(d)
        streams.append(bad); sizes.append(n if 0 < n <= (16 << 20) else len(d))
    good, kw = random_plain(3)
    streams.append(E.encode(good, **kw)); sizes.append(len(good))
    sim.sim_set_duo(duo)
    try:
        outs, status = run_batch(sim, streams, sizes, grid=64)
    finally:
        sim.sim_set_duo(0)
    assert np.array_equal(outs[-1], good)
