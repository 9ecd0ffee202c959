#!/usr/bin/env python3
"""Round 6 diagnostics: the workspace header words after a decode of a bench-shaped batch (status, policy, totals), for the library named by
BROTLIG_HIP_SO (default: the in-tree one), and the phase profile of the same batch."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from brotli_g_sdk_amd import api
kind = sys.argv[1] if len(sys.argv) > 1 else "mixed"
nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 16
streams, _ = bench.build_streams(kind, range(nstreams), 4096, 256)
dec = api.BatchDecoder(streams)
dec.decode()
torch.cuda.synchronize()
words = dec.d_ws[:160 * 4].cpu().numpy().view("uint32")
tot, kern = dec.timed(2, 5)
p = dec.phase_profile()
t = p["total"]
print(json.dumps({"lib": os.environ.get("BROTLIG_HIP_SO", "in-tree"), "workload": kind, "status": int(words[0]), "counter": int(words[1]), "precon": int(words[2]), "policy": int(words[3]),
                  "supers": int(words[5]), "pages_word6": int(words[6]), "hist": [int(x) for x in words[8:72]], "kernel_ms": round(kern, 4), "step_ms": round(tot / 5, 4),
                  "setup": round(p["setup"] / t, 4), "tables": round(p["tables"] / t, 4), "commands": round(p["commands"] / t, 4), "rounds": p["rounds"], "solo_rounds": p["solo_rounds"],
                  "cycles_per_round": round(t / max(p["rounds"], 1), 1), "total_cycles": t}))
