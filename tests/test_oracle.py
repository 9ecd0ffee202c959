"""CPU tests of the oracle (oracle/brotlig_oracle.c) and of the encoder it is exercised with.

The reference ships no tests, fixtures or sample streams, so there is no reference
golden vector to pin the oracle to; what is checked here is (a) encoder -> oracle round trips over
every bitstream feature the format has, (b) committed regression fixtures under tests/golden/
(made by tests/golden/make_golden.py with this repo's encoder and oracle -- NOT reference output),
(c) the reference's two header checks."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from brotli_g_sdk_amd import encoder as E
from cases import plain_cases, precon_cases
from helpers import oracle_decode, oracle_lib

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name,thunk,kw", plain_cases(), ids=[c[0] for c in plain_cases()])
def test_roundtrip_plain(name, thunk, kw):
    data = thunk()
    stream = E.encode(data, **kw)
    rc, out = oracle_decode(stream)
    assert rc == 0
    assert len(out) == len(data) and np.array_equal(out, data)


@pytest.mark.parametrize("name,thunk,pre", precon_cases(), ids=[c[0] for c in precon_cases()])
def test_roundtrip_preconditioned(name, thunk, pre):
    tex = thunk()
    stream = E.encode(tex, precondition=pre)
    rc, out = oracle_decode(stream, out_size=len(tex))
    assert rc == 0
    assert np.array_equal(out, tex)


def test_config1_is_a_stored_page():
    """BASELINE config 1: one 64 KiB page of random bytes -> 8 + 4 + 65536 byte stream."""
    data = np.random.default_rng(0).integers(0, 256, 65536, dtype=np.uint8)
    stream = E.encode(data)
    assert len(stream) == 8 + 4 + 65536
    assert bytes(stream[:8]) == bytes([5, 0xFA, 1, 0, 1, 0, 0, 0])
    assert oracle_lib().DecompressedSize(stream.ctypes.data) == 65536
    rc, out = oracle_decode(stream)
    assert rc == 0 and np.array_equal(out, data)


def test_header_bytes_match_survey_example():
    """three stored pages (2 x 65536 + 1000) give header 05 fa 03 00 a1 0f 00 00."""
    data = np.arange(2 * 65536 + 1000, dtype=np.uint32).astype(np.uint8)
    stream = E.encode(data, flags=E.FORCE_STORED)
    assert bytes(stream[:8]) == bytes.fromhex("05fa0300a10f0000")
    assert oracle_lib().DecompressedSize(stream.ctypes.data) == len(data)


def test_error_codes():
    """src/BrotligDecoder.cpp:437-446: magic mismatch -> CORRUPT_STREAM (14), id != 5 -> INCORRECT_STREAM_FORMAT (15)."""
    stream = E.encode(np.zeros(1000, np.uint8))
    bad_magic = stream.copy(); bad_magic[1] ^= 0x10
    rc, _ = oracle_decode(bad_magic, out_size=1000)
    assert rc == 14
    bad_id = stream.copy(); bad_id[0] = 6; bad_id[1] = 6 ^ 0xFF
    rc, _ = oracle_decode(bad_id, out_size=1000)
    assert rc == 15


def test_worker_policy():
    """src/BrotligDecoder.cpp:404-415: more than 2*workers pages -> min(128, hw threads) workers, else one."""
    L = oracle_lib()
    hw = min(128, os.cpu_count() or 1)
    small = E.encode(np.zeros(2 * 65536, np.uint8))
    big = E.encode(np.zeros((2 * hw + 1) * 65536, np.uint8))
    for stream, expect in ((small, 1), (big, hw)):
        n = L.DecompressedSize(stream.ctypes.data)
        out = np.empty(n, np.uint8)
        osz, used = ctypes.c_uint32(n), ctypes.c_int(0)
        assert L.brotlig_oracle_decode(len(stream), stream.ctypes.data, ctypes.byref(osz), out.ctypes.data, 0, ctypes.byref(used)) == 0
        assert used.value == expect
        assert not out.any()


def test_golden_fixtures():
    index = json.load(open(os.path.join(GOLDEN, "index.json")))
    assert len(index) >= 12
    for name, meta in index.items():
        stream = np.fromfile(os.path.join(GOLDEN, name + ".brotlig"), dtype=np.uint8)
        rc, out = oracle_decode(stream, out_size=meta["size"])
        assert rc == 0, name
        assert hashlib.sha256(out.tobytes()).hexdigest() == meta["sha256"], name
