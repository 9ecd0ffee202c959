#!/usr/bin/env python3
"""Round 5: what a batch of MANY small streams costs (an asset streamer's hand-over): step time against decode-kernel time for
n streams of 1 byte .. a page, n = 256 .. 65 536.  The difference is the prepare kernel (one workgroup walks the headers, 32 per
step), the schedule kernels and the launches.  Prints one JSON line per n."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from brotli_g_sdk_amd import api, datagen as D, encoder as E   # noqa: E402

rng = np.random.default_rng(91)
makers = [D.text, D.records, D.samples16, D.runs, D.mixed, D.random_bytes]
base = [makers[i % 6](int(rng.integers(1, 65536 + 3000)) if i % 5 else int(rng.integers(1, 40)), 3000 + i) for i in range(96)]
enc = [E.encode(d) for d in base]
for n in (256, 2048, 20011, 65536):
    pick = rng.integers(0, 96, n)
    dec = api.BatchDecoder([enc[k] for k in pick], out_sizes=[len(base[k]) for k in pick])
    dec.decode()
    total, kern = dec.timed(3, 20)
    out_bytes = int(sum(len(base[k]) for k in pick))
    import time
    dec.stream_status()
    t0 = time.perf_counter()
    for _ in range(5):
        dec.stream_status()
    status_ms = (time.perf_counter() - t0) / 5 * 1e3
    print(json.dumps({"streams": n, "decompressed_MB": round(out_bytes / 1e6, 1), "step_ms": round(total / 20, 4), "decode_kernel_ms": round(kern, 4),
                      "outside_kernel_ms": round(total / 20 - kern, 4), "GBps_step": round(out_bytes / (total / 20) / 1e6, 1),
                      "stream_status_call_ms": round(status_ms, 3)}), flush=True)
    del dec
