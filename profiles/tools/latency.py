#!/usr/bin/env python3
"""Small-batch behaviour: kernel time of 1, 8, 64, 512 and 4096 pages (config 2 data and mixed data)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from brotli_g_sdk_amd import api, datagen as D, encoder as E
out = {}
for kind in ("runs", "mixed"):
    for pages in (1, 8, 64, 512, 4096):
        base = (D.runs if kind == "runs" else D.mixed)(min(pages, 256) * 65536, 1)
        s = E.encode(base)
        if pages > 256: s = D.tile_stream(s, pages // 256)
        dec = api.BatchDecoder([s])
        dec.decode()
        tot, k = dec.timed(3, 20)
        assert np.array_equal(dec.output(0)[:len(base)], base)
        out[f"{kind}_{pages}"] = {"kernel_ms": round(k, 4), "GBps": round(pages * 65536 / k / 1e6, 1)}
print(json.dumps(out))
