#!/usr/bin/env python3
"""Round 5 probe: what a heavy page costs alone in a wavefront, next to a light page, and next to another heavy one (classic kernel pinned)."""
import json, os, sys
os.environ.setdefault("BROTLIG_ENABLE_DEBUG_KNOBS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from brotli_g_sdk_amd import api, datagen as D, encoder as E   # noqa: E402

def enc(fmt, w, h, mips, seed):
    tex = D.bc_texture(fmt, w, h, seed=seed, num_mips=mips)
    return E.encode(tex, precondition=dict(format=fmt, width_blocks=w, height_blocks=h, num_mips=mips, swizzle=1, delta=1)), len(tex)

api.DebugSetDecodeMode(1)
cases = {
    "bc5 64x64 1 mip: one full page alone": [enc(5, 64, 64, 1, 1)],
    "bc5 64x64 7 mips: a full page and a 23 KiB page in one wavefront": [enc(5, 64, 64, 7, 2)],
    "two bc5 64x64 1 mip: two full pages in one wavefront": [enc(5, 64, 64, 1, 3), enc(5, 64, 64, 1, 4)],
    "bc3 64x64 1 mip alone": [enc(3, 64, 64, 1, 5)],
    "two bc3 64x64 1 mip in one wavefront": [enc(3, 64, 64, 1, 6), enc(3, 64, 64, 1, 7)],
    "bc1 64x64 7 mips (one 44 KiB page) alone": [enc(1, 64, 64, 7, 8)],
}
for name, items in cases.items():
    for grid in (1, 0):
        api.DebugSetDecodeGrid(grid)
        dec = api.BatchDecoder([s for s, _ in items], out_sizes=[n for _, n in items])
        dec.decode()
        total, kern = dec.timed(3, 20)
        print(json.dumps({"case": name, "wavefronts": "one" if grid else "as the host chooses", "decode_kernel_ms": round(kern, 4)}), flush=True)
        del dec
