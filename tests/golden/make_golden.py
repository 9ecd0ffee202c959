#!/usr/bin/env python3
"""Regenerates tests/golden/: small .brotlig streams plus the SHA-256 of their decoded bytes.

Provenance, stated plainly: streams come from this repo's encoder and the expected digests from
this repo's oracle (oracle/brotlig_oracle.c).  They are regression vectors, not reference output:
the reference has no fixtures and cannot be built under this repo's rules (DESIGN.md, section
"Oracle").  The digest of each case also equals the digest of the bytes that were encoded."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from brotli_g_sdk_amd import datagen as D, encoder as E  # noqa: E402
from cases import many_command_shapes, many_distances, skewed  # noqa: E402
from helpers import oracle_decode  # noqa: E402

N = 65536 + 9000
CASES = {
    "stored_random": (D.random_bytes(40000, 0), {}),
    "zeros": (np.zeros(70000, np.uint8), {}),
    "runs": (D.runs(N, 1), {}),
    "text": (D.text(N, 2), {}),
    "records_np2": (D.records(N, 3), dict(npostfix=2, ndirect_m=5)),
    "samples16": (D.samples16(N, 4), {}),
    "text_no_rle": (D.text(N, 5), dict(flags=E.NO_CODELEN_RLE)),
    "records_no_ring": (D.records(N, 6), dict(flags=E.NO_RING_CODES)),
    "literals_only": (D.text(40000, 7), dict(flags=E.LITERALS_ONLY)),
    "skewed_long_codes": (skewed(N, 8), {}),
    "text_32k": (D.text(N, 9), dict(page_size=32768)),
    "mixed_128k": (D.mixed(2 * 65536, 10), dict(page_size=131072)),
    "bc1_swz_delta": (D.bc_texture(1, 64, 64, seed=11), dict(precondition=dict(format=1, width_blocks=64, height_blocks=64, swizzle=1, delta=1))),
    "bc3_mips": (D.bc_texture(3, 40, 24, seed=12, num_mips=3), dict(precondition=dict(format=3, width_blocks=40, height_blocks=24, num_mips=3, swizzle=1, delta=1))),
    # more ICP / distance symbols in a page than the decoder's LDS arrays hold (kIcpSymCap, kDistSymCap)
    "many_command_shapes_128k": (many_command_shapes(131072 + 500, 5), dict(page_size=131072)),
    "many_distances_np3": (many_distances(65536 + 500, 6), dict(npostfix=3, ndirect_m=15)),
    "bc5_pitch": (D.bc_texture(5, 33, 17, seed=13, pitch_bytes=33 * 16 + 7), dict(precondition=dict(format=5, width_blocks=33, height_blocks=17, swizzle=1, delta=1, pitch_bytes=33 * 16 + 7))),
}

index = {}
for name, (data, kw) in CASES.items():
    stream = E.encode(data, **kw)
    rc, out = oracle_decode(stream, out_size=len(data))
    assert rc == 0 and np.array_equal(out, data), name
    stream.tofile(os.path.join(HERE, name + ".brotlig"))
    index[name] = {"size": int(len(data)), "sha256": hashlib.sha256(out.tobytes()).hexdigest(),
                   "preconditioned": "precondition" in kw}
json.dump(index, open(os.path.join(HERE, "index.json"), "w"), indent=1, sort_keys=True)
print(f"wrote {len(index)} fixtures, {sum(os.path.getsize(os.path.join(HERE, n + '.brotlig')) for n in index)} bytes")
