#!/usr/bin/env python3
"""Small-batch behaviour: kernel time of 1, 8, 64, 512 and 4096 pages (config 2 data and mixed data)."""
import os as _os; _os.environ.setdefault("BROTLIG_ENABLE_DEBUG_KNOBS", "1")    # the kernel-selection switches are inert without it (diagnostics only)
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from brotli_g_sdk_amd import api, datagen as D, encoder as E
out = {}
MODES = {"auto": 0, "one_wavefront": 1, "two_wavefronts": 2}
modes = [m for m in sys.argv[1:] if m in MODES] or ["auto"]
for mode in modes:
  api.DebugSetDecodeMode(MODES[mode])
  tag = "" if mode == "auto" else "_" + mode
  for kind in ("runs", "mixed"):
    for pages in (1, 8, 64, 512, 1024, 2048, 4096):
        base = (D.runs if kind == "runs" else D.mixed)(min(pages, 256) * 65536, 1)
        s = E.encode(base)
        if pages > 256: s = D.tile_stream(s, pages // 256)
        dec = api.BatchDecoder([s])
        dec.decode()
        tot, k = dec.timed(3, 20)
        assert np.array_equal(dec.output(0)[:len(base)], base)
        out[f"{kind}_{pages}{tag}"] = {"kernel_ms": round(k, 4), "GBps": round(pages * 65536 / k / 1e6, 1)}
api.DebugSetDecodeMode(0)
# single assets through the host-pointer entry: DecodeGPU (a context per call: allocations) against a reusable BrotligContext
import time
ctx = api.Context()
for name, n in (("asset_64KiB", 65536), ("asset_1MiB", 1 << 20), ("asset_16MiB", 16 << 20)):
    data = D.mixed(n, 5)
    s = E.encode(data)
    for label, fn in (("DecodeGPU", api.DecodeGPU), ("BrotligContextDecodeGPU", ctx.DecodeGPU)):
        o, k = fn(s)
        assert np.array_equal(o, data)
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < 0.5:
            o, k = fn(s); reps += 1
        wall = (time.perf_counter() - t0) / reps * 1e3
        out[f"{name}_{label}"] = {"wall_ms_host_to_host": round(wall, 3), "kernel_ms": round(k, 4), "GBps_host_to_host": round(n / wall / 1e6, 2)}
ctx.close()
print(json.dumps(out))
