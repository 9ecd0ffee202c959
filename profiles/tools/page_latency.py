#!/usr/bin/env python3
"""Latency of ONE 64 KiB page per data class (the small-batch regime: one page per wavefront, whole SIMD to itself), with the
phase shares of that launch: which classes decide the 512-page / config-2 numbers, and where their time goes."""
import os as _os; _os.environ.setdefault("BROTLIG_ENABLE_DEBUG_KNOBS", "1")    # the kernel-selection switches are inert without it (diagnostics only)
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from brotli_g_sdk_amd import api, datagen as D, encoder as E
MODE = {"auto": 0, "one_wavefront": 1, "two_wavefronts": 2}[sys.argv[1] if len(sys.argv) > 1 else "auto"]
api.DebugSetDecodeMode(MODE)
COUNTS = ("rounds", "levels", "solo_rounds", "groups", "lit_steps", "team_levels", "level_halves", "group_halves")
for kind in ("text", "records", "samples16", "runs", "random_bytes"):
    base = getattr(D, kind)(65536, 7)
    s = E.encode(base)
    dec = api.BatchDecoder([s])
    dec.decode()
    _, k = dec.timed(3, 20)
    p = dec.phase_profile()
    tot = p["total"]
    out = {"mode": MODE, "class": kind, "compressed": len(s), "kernel_ms": round(k, 4)}
    out.update({n: (v if n in COUNTS else round(v / tot, 4)) for n, v in p.items()})
    out["cycles_per_round"] = round(tot / max(p["rounds"], 1), 1)
    print(json.dumps(out), flush=True)
