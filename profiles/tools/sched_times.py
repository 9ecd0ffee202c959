#!/usr/bin/env python3
"""Round 6 diagnostics: where the schedule kernel (the one launch in front of the page decode, csrc/brotlig_schedule.h) spends its time.
Four ticks of the 100 MHz counter per ticket -- came, got the ticket, saw its phase open, done -- grouped by phase.
  python profiles/tools/sched_times.py [workload[:streams[:pages]]] ..."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from brotli_g_sdk_amd import api

for spec in (sys.argv[1:] or ["mixed"]):
    w, *shape = spec.split(":")
    nstreams = int(shape[0]) if shape else 16
    npages = int(shape[1]) if len(shape) > 1 else 4096
    streams, expected = bench.build_streams(w, list(range(nstreams)), npages, min(256, npages))
    dec = api.BatchDecoder(streams, out_sizes=[len(e) for e in expected] if w == "bc3" else None)
    dec.decode()
    dec.schedule_times()                      # (once warm)
    t = dec.schedule_times().astype(np.int64)
    published = int(t[-1, 3])                 # (the scatter item that finished last: when it had published the batch)
    t = t[:-1]
    live = t[:, 3] > 0
    n = int(live.sum())
    if n == 0:
        print(json.dumps({"workload": spec, "tickets": 0, "note": "one workgroup ran the phases (a small batch): no tickets"}))
        continue
    t = t[live]
    t0 = int(t[:, 0].min())
    us = lambda x: round(float(x) / 100.0, 2)
    # phases by ticket ranges: reconstructed from the gaps is fragile; report quantiles over tickets instead, in ticket order blocks
    pages = int(sum(api.DecompressedSize(s) for s in streams) // 65536)
    out = {"workload": spec, "pages": pages, "tickets": n, "first_came_to_last_done_us": us(t[:, 3].max() - t0), "published_us": us(published - t0),
           "came_spread_us": us(t[:, 0].max() - t0), "ticket_latency_us_median": us(np.median(t[:, 1] - t[:, 0])),
           "by_ticket_block": []}
    nblk = 8
    for b in range(nblk):
        blk = t[b * n // nblk:(b + 1) * n // nblk]
        if len(blk):
            out["by_ticket_block"].append({"tickets": [b * n // nblk, (b + 1) * n // nblk - 1], "came_us": us(np.median(blk[:, 0]) - t0),
                                           "open_us": us(np.median(blk[:, 2]) - t0), "done_us": us(np.median(blk[:, 3]) - t0),
                                           "wait_us": us(np.median(blk[:, 2] - blk[:, 1])), "work_us": us(np.median(blk[:, 3] - blk[:, 2]))})
    out["ticket0"] = {"came": 0.0 if int(t[0, 0]) == t0 else us(t[0, 0] - t0), "open": us(t[0, 2] - t0), "done": us(t[0, 3] - t0)}
    out["last_ticket"] = {"came": us(t[-1, 0] - t0), "open": us(t[-1, 2] - t0), "done": us(t[-1, 3] - t0)}
    print(json.dumps(out))
