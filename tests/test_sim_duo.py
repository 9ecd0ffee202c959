"""The small-batch kernel -- two wavefronts per page, brotlig_decode_duo_kernel (brotli_g_sdk_amd/csrc/brotlig_kernels.h) -- on the CPU
simulator: the cases of tests/test_sim_decode.py and tests/test_sim_fuzz.py once more, every launch through the producer / consumer pair
(two wave64s per workgroup, handed over through the LDS step ring; the simulator runs the wavefronts of a workgroup interleaved and
lets a polling wavefront yield, tests/sim/sim_runtime.h).  The device runs the same source: tests/test_gpu_decode.py (-m gpu)."""
import ctypes

import numpy as np
import pytest

from brotli_g_sdk_amd import datagen as D
from brotli_g_sdk_amd import encoder as E
from helpers import oracle_decode
from test_sim_decode import build_sim, run_batch
from test_sim_decode import (test_sim_plain, test_sim_preconditioned, test_sim_batch_of_streams, test_sim_far_boundary,  # noqa: F401
                             test_sim_far_boundaries_one_page_per_wavefront, test_sim_symbol_overflow, test_sim_bad_header_sets_status,
                             test_sim_simple_code_with_one_symbol_rejects_the_page)
from test_sim_fuzz import (test_sim_fuzz, test_sim_corrupt_streams_terminate, test_sim_random_options, test_sim_random_precondition,  # noqa: F401
                           test_sim_corrupt_variants_stay_in_bounds, test_sim_damaged_header_cannot_reach_a_neighbouring_stream)


@pytest.fixture(scope="module")
def sim():
    L = build_sim("libbrotlig_sim.so")
    L.sim_duo_launches.restype = ctypes.c_uint64
    return L


@pytest.fixture(autouse=True)
def two_wavefronts_per_page(sim):
    """Every decode of this module goes through brotlig_decode_duo_kernel -- and is seen to (the launch counter moves)."""
    before = sim.sim_duo_launches()
    sim.sim_set_duo(1)
    yield
    sim.sim_set_duo(0)
    assert sim.sim_duo_launches() > before


def test_sim_duo_more_pages_than_workgroups_and_stored_pages_between(sim):
    """Two workgroups, eleven pages of four kinds -- compressed, stored (random bytes: copied by the producer on the spot, no step
    record), a short last page: each producer / consumer pair takes page after page from the counter, the ring carries page start,
    groups, page end from one page to the next, and the pair leaves together when the counter runs out."""
    parts = [D.text(65536, 1), D.random_bytes(65536, 2), D.runs(65536, 3), D.records(65536, 4), D.random_bytes(65536, 5),
             D.samples16(65536, 6), D.text(65536, 7), D.runs(2 * 65536, 8), D.mixed(65536 + 777, 9)]
    data = np.concatenate(parts)
    stream = E.encode(data)
    rc, ref = oracle_decode(stream)
    assert rc == 0 and np.array_equal(ref, data)
    for grid in (1, 2, 16):
        outs, status = run_batch(sim, [stream], [len(data)], grid=grid)
        assert status == 0 and np.array_equal(outs[0], ref), grid


def test_sim_duo_rounds_of_many_groups(sim):
    """Rounds far longer than a group (1 024 bytes): long inserts and long copies, so that one round is many steps of the ring (more
    than its four slots: the producer has to wait for the consumer) and the literal carry crosses group boundaries."""
    rng = np.random.default_rng(11)
    noise = rng.integers(0, 256, 9000, dtype=np.uint8)
    data = np.concatenate([noise, np.tile(noise[:4000], 6), rng.integers(0, 4, 7000, dtype=np.uint8).astype(np.uint8), noise[1000:8000],
                           np.zeros(12000, np.uint8), noise])
    stream = E.encode(data)
    rc, ref = oracle_decode(stream)
    assert rc == 0 and np.array_equal(ref, data)
    outs, status = run_batch(sim, [stream], [len(data)], grid=2)
    assert status == 0 and np.array_equal(outs[0], ref)


@pytest.mark.parametrize("seed", [150069, 150248, 150466, 151219, 151529])
def test_sim_duo_damaged_distance_below_zero_is_not_copied(sim, seed):
    """Damaged streams from the device soak (profiles/tools/soak.py, seeds 150000..): a ring code applied to a small distance wraps below
    zero (1 - 3).  decode_pages refuses the copy (distance > position); the producer used to hand the raw distance to the consumer with
    its flags OR-ed on top, the wrapped value's high bits read as "valid copy", and the consumer fetched from far outside the buffers (a
    memory access fault on the device, a segmentation fault here).  The page must end with the bad-page status and nothing else."""
    from fuzzcases import corrupt, random_plain
    d, kw = random_plain(seed)
    bad, kind = corrupt(E.encode(d, **kw), seed)
    outs, status = run_batch(sim, [bad], [len(d)], grid=2)
    assert status & 2, (seed, kind, status)


def _damaged_batch_150080():
    """Batch 150080 of the device soak: forty damaged streams and a valid one, output sizes as the damaged headers claim them."""
    from brotli_g_sdk_amd.api import DecompressedSize
    from fuzzcases import corrupt, random_plain
    streams, sizes = [], []
    for seed in range(150080, 150120):
        d, kw = random_plain(seed)
        bad, kind = corrupt(E.encode(d, **kw), seed)
        try:
            n = int(DecompressedSize(bad))
        except Exception:
            n = len(d)
        streams.append(bad); sizes.append(n if n <= (64 << 20) else len(d))
    good, kw = random_plain(3)
    streams.append(E.encode(good, **kw)); sizes.append(len(good))
    return streams, sizes, good


def test_sim_empty_page_behind_a_damaged_table_entry_is_rejected(sim):
    """A damaged page table can describe a page of zero bytes anywhere: beyond its stream the room is 0, and "0 > 0" let it through
    fetch_job -- the bit readers then started at an address outside the input (a memory access fault on the device in round 4's soak, in
    either kernel; a segmentation fault here).  An empty page is rejected now; the valid stream next to it stays bit-exact."""
    streams, sizes, good = _damaged_batch_150080()
    outs, status = run_batch(sim, streams, sizes, grid=300)
    assert status != 0 and np.array_equal(outs[-1], good)
    sim.sim_set_duo(0)                                              # and through the one-wavefront kernel
    outs, status = run_batch(sim, streams, sizes, grid=300)
    sim.sim_set_duo(1)
    assert status != 0 and np.array_equal(outs[-1], good)
