#!/usr/bin/env python3
"""Diagnostics: per-phase cycle breakdown of the page-decode kernel on a bench-shaped batch."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from brotli_g_sdk_amd import api
bench.ENCODER_FLAGS = int(os.environ.get("BROTLIG_ENCODER_FLAGS", "0"))      # 192: the optimal-parse streams bench.py reports as `alt`
kind = sys.argv[1] if len(sys.argv) > 1 else "mixed"
nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 4
streams, _ = bench.build_streams(kind, range(nstreams), 2048, 128)
dec = api.BatchDecoder(streams)
dec.decode()
p = dec.phase_profile()
tot = p["total"]
out = {k: (v if k in ("rounds", "levels", "solo_rounds", "groups", "lit_steps", "team_levels", "level_halves", "group_halves") else round(v / tot, 4)) for k, v in p.items()}
out["levels_per_round"] = round(p["levels"] / max(p["rounds"], 1), 2)
# lock-step cost: of the two halves of a wavefront, how many had work in an iteration of a loop that runs for both
out["halves_per_level"] = round(p["level_halves"] / max(p["levels"], 1), 3)
out["halves_per_group"] = round(p["group_halves"] / max(p["groups"], 1), 3)
out["cycles_per_round"] = round(tot / max(p["rounds"], 1), 1)
out["pages"] = int(sum(api.DecompressedSize(s) for s in streams) // 65536)
out["workload"] = kind + ("" if not bench.ENCODER_FLAGS else f" (encoder flags {bench.ENCODER_FLAGS})")
print(json.dumps(out))
