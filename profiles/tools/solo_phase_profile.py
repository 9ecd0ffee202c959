#!/usr/bin/env python3
"""Phase shares of the decode kernel in the small-batch regime (one page per wavefront, upper half idle): config 2 (one stream
of 4096 'runs' pages), 512 mixed pages, one mixed page."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from brotli_g_sdk_amd import api, datagen as D, encoder as E
for kind, pages in (("runs", 4096), ("mixed", 512), ("mixed", 1), ("text", 256)):
    base = (D.runs if kind == "runs" else D.mixed if kind == "mixed" else D.text)(min(pages, 256) * 65536, 1)
    s = E.encode(base)
    if pages > 256:
        s = D.tile_stream(s, pages // 256)
    dec = api.BatchDecoder([s])
    dec.decode()
    _, k = dec.timed(2, 5)
    p = dec.phase_profile()
    tot = p["total"]
    out = {"workload": kind, "pages": pages, "kernel_ms": round(k, 4)}
    out.update({n: (v if n in ("rounds", "levels", "solo_rounds", "groups", "lit_steps", "team_levels", "level_halves", "group_halves") else round(v / tot, 4)) for n, v in p.items()})
    out["cycles_per_round"] = round(tot / max(p["rounds"], 1), 1)
    print(json.dumps(out))
