#!/usr/bin/env python3
"""Soak run on the MI355X box: the generators of tests/test_gpu_differential.py over many more seeds (plain and
pre-conditioned streams against the bytes they were encoded from, damaged streams next to valid ones).
  python profiles/tools/soak.py [first_seed] [n_plain] [n_precon] [n_corrupt] [mode]
mode: 0 = the kernel the batch size selects (these batches: two wavefronts per page), 1 = one wavefront per one or two pages, 2 = two wavefronts per page."""
import os as _os; _os.environ.setdefault("BROTLIG_ENABLE_DEBUG_KNOBS", "1")    # the kernel-selection switches are inert without it (diagnostics only)
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from brotli_g_sdk_amd import api, encoder as E
from fuzzcases import corrupt, random_plain, random_precon
from cases import symbol_overflow_cases

first = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n_plain = int(sys.argv[2]) if len(sys.argv) > 2 else 1600
n_precon = int(sys.argv[3]) if len(sys.argv) > 3 else 300
n_corrupt = int(sys.argv[4]) if len(sys.argv) > 4 else 400
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0
api.DebugSetDecodeMode(mode)
grid = int(os.environ.get("BROTLIG_SOAK_GRID", "0"))     # e.g. 2: the classic kernel on two wavefronts -- every batch two pages per wavefront, the form of the large batches
if grid: api.DebugSetDecodeGrid(grid)
t0 = time.time(); bad = []
B = 80
for c in range(first, first + n_plain, B):
    print('plain', c, file=sys.stderr, flush=True)      # (the last line before a fault names the batch)
    items = [random_plain(s) for s in range(c, min(c + B, first + n_plain))]
    streams = [E.encode(d, **kw) for d, kw in items]
    dec = api.BatchDecoder(streams); dec.poison_output(); dec.decode()
    for i, (d, kw) in enumerate(items):
        if not np.array_equal(dec.output(i), d): bad.append(("plain", c + i))
for c in range(first, first + n_precon, B):
    print('precon', c, file=sys.stderr, flush=True)
    items = [random_precon(s) for s in range(c, min(c + B, first + n_precon))]
    streams = [E.encode(t, precondition=pre, **kw) for t, pre, kw in items]
    dec = api.BatchDecoder(streams, out_sizes=[len(t) for t, _, _ in items]); dec.poison_output(); dec.decode()
    for i, (t, pre, kw) in enumerate(items):
        # the encoder's input is the reference here: a texture round-trips exactly except for row-pitch padding (zero on output)
        out = dec.output(i)
        if len(out) != len(t) or (pre.get("pitch_bytes", 0) == 0 and not np.array_equal(out, t)): bad.append(("precon", c + i))
valid = [E.encode(*random_plain(3)[:1], **random_plain(3)[1])]
vref = random_plain(3)[0]
statuses = {}
for c in range(first, first + n_corrupt, 40):
    print('corrupt', c, file=sys.stderr, flush=True)
    streams, sizes = [], []
    for s in range(c, min(c + 40, first + n_corrupt)):
        d, kw = random_plain(s)
        st = E.encode(d, **kw)
        b, kind = corrupt(st, s)
        streams.append(b); sizes.append(len(d))
    streams.append(valid[0]); sizes.append(len(vref))
    try:
        dec = api.BatchDecoder(streams, out_sizes=None)
        dec.poison_output()
        try:
            dec.decode(); statuses["ok"] = statuses.get("ok", 0) + 1
        except api.BrotligError as e:
            statuses[e.code] = statuses.get(e.code, 0) + 1
        if not np.array_equal(dec.output(len(streams) - 1), vref): bad.append(("valid-next-to-corrupt", c))
    except api.BrotligError as e:
        statuses["refused"] = statuses.get("refused", 0) + 1
    out, _ = api.DecodeGPU(valid[0])
    if not np.array_equal(out, vref): bad.append(("valid-after-corrupt", c))
print(json.dumps({"decode_mode": mode, "forced_grid": grid, "first_seed": first, "plain": n_plain, "precon": n_precon, "corrupt": n_corrupt, "failures": bad[:20], "n_failures": len(bad),
                  "corrupt_batch_statuses": {str(k): v for k, v in statuses.items()}, "seconds": round(time.time() - t0, 1)}))
