"""Seeded fuzz of the kernel source on the CPU simulator: random compositions of literal runs,
short/long repeats at near and far distances, byte runs and self-overlapping copies, so that rounds
alternate between the windowed path and the global-memory path inside one page."""
import numpy as np
import pytest

from brotli_g_sdk_amd import encoder as E
from helpers import oracle_decode
from test_sim_decode import run_batch, sim  # noqa: F401  (fixture)


def compose(seed, n):
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
            d = int(rng.integers(1, min(pos, 1500)) ) if pos > 1 else 1
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
