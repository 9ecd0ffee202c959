"""Round 6: the real-bytes source (datagen.files -- files the image ships) on the CPU side: the sample is what its manifest says, every kind
contributes, and the oracle and the CPU library decode its streams back bit for bit.  (The GPU twin: tests/test_gpu_fullsize.py.)"""
import numpy as np
import pytest

from brotli_g_sdk_amd import datagen as D
from brotli_g_sdk_amd import encoder as E
from helpers import oracle_decode


def _have_files():
    try:
        D.files(4096, 0)
        return True
    except RuntimeError:
        return False


pytestmark = pytest.mark.skipif(not _have_files(), reason="none of datagen.FILE_ROOTS exists on this machine")


def test_files_sample_is_reproducible_and_mixed():
    m1, m2 = [], []
    a, b = D.files(6 << 20, 3, manifest=m1), D.files(6 << 20, 3, manifest=m2)
    assert np.array_equal(a, b) and m1 == m2
    assert not np.array_equal(a, D.files(6 << 20, 4))
    s = D.files_manifest_summary(m1)
    assert sum(v["bytes"] for v in s["kinds"].values()) == 6 << 20
    assert len(s["kinds"]) >= 3 and min(v["bytes"] for v in s["kinds"].values()) > (6 << 20) // 8


@pytest.mark.parametrize("flags", [0, E.OPTIMAL_PARSE | E.SEARCH_DIST_PARAMS], ids=["lazy", "optimal"])
def test_files_round_trip_through_the_oracle(flags):
    d = D.files(3 * 65536 + 4321, 1)
    s = E.encode(d, flags=flags)
    assert len(s) < len(d)
    rc, out = oracle_decode(s)
    assert rc == 0 and np.array_equal(out, d)
    from brotli_g_sdk_amd import cpu
    rc, out = cpu.DecodeCPU(s)
    assert rc == 0 and np.array_equal(out, d)
