#!/usr/bin/env python3
"""Round 5: batches of MANY SMALL pre-conditioned textures (what texture streaming hands over), one batch each: step time against the
decode-kernel time -- the difference is the prepare kernels and the de-conditioning pass -- and the rate of that pass by its bytes
(conditioned bytes read + texture bytes written).  Prints one JSON line per batch shape."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from brotli_g_sdk_amd import api, datagen as D, encoder as E   # noqa: E402

SHAPES = [  # (label, count, [(format, width, height, mips)] cycled)
    ("4096 x BC3 64x64 blocks (64 KiB), 1 mip", 4096, [(3, 64, 64, 1)]),
    ("4096 x BC1/BC3/BC5 64x64 blocks, 7 mips", 4096, [(1, 64, 64, 7), (3, 64, 64, 7), (5, 64, 64, 7)]),
    ("1024 x BC3 256x256 blocks (1 MiB), 9 mips", 1024, [(3, 256, 256, 9)]),
    ("mixed: 1 in 64 BC3 1024x1024 (16 MiB), the rest BC1 32x32..128x128, full mip chains", 2048,
     [(3, 1024, 1024, 11)] + [(1, 32, 32, 6), (1, 64, 64, 7), (1, 128, 128, 8)] * 21),
    ("256 x BC3 1024x1024 blocks (16 MiB), 1 mip: config 4", 256, [(3, 1024, 1024, 1)]),
]
for label, count, kinds in SHAPES:
    enc = {}
    for k, (fmt, w, h, mips) in enumerate(kinds):
        if (fmt, w, h, mips) in enc:
            continue
        tex = D.bc_texture(fmt, w, h, seed=500 + k, num_mips=mips)
        pre = dict(format=fmt, width_blocks=w, height_blocks=h, num_mips=mips, swizzle=1, delta=1)
        enc[(fmt, w, h, mips)] = (E.encode(tex, precondition=pre), len(tex))
    pick = [kinds[i % len(kinds)] for i in range(count)]
    dec = api.BatchDecoder([enc[k][0] for k in pick], out_sizes=[enc[k][1] for k in pick])
    dec.decode()
    total, kern = dec.timed(3, 10)
    out_bytes = int(sum(enc[k][1] for k in pick))
    rest = total / 10 - kern
    print(json.dumps({"batch": label, "textures": count, "texture_MB": round(out_bytes / 1e6, 1), "step_ms": round(total / 10, 4), "decode_kernel_ms": round(kern, 4),
                      "prepare_and_decondition_ms": round(rest, 4), "decondition_TBps_at_most": round(2 * out_bytes / rest / 1e9, 2),
                      "GBps_step": round(out_bytes / (total / 10) / 1e6, 1)}), flush=True)
    del dec
