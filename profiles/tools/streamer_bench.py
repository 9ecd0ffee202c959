#!/usr/bin/env python3
"""Host-to-host throughput of the streaming front end (BrotligStreamer*): compressed streams in host
memory in, decoded bytes in (pinned) host memory out, PCIe both ways.  Prints one JSON line.
Usage: python profiles/tools/streamer_bench.py [batches=12] [streams_per_batch=4] [pages_per_stream=1024]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from brotli_g_sdk_amd import api

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 12
spb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pps = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
streams, expected = bench.build_streams("mixed", range(spb), pps, 128)
c_bytes = sum(len(s) for s in streams)
u_bytes = sum(api.DecompressedSize(s) for s in streams)
res = {"workload": f"{batches} batches x {spb} mixed streams x {pps} pages ({u_bytes >> 20} MiB decoded, {c_bytes >> 20} MiB compressed per batch)"}
for slots in (1, 3):
    st = api.Streamer(slots=slots, slot_in_bytes=c_bytes + (1 << 20), slot_out_bytes=u_bytes + (1 << 20), max_streams=spb)
    t = st.submit(streams); outs = st.result(t)                     # warm-up + correctness
    for o, e in zip(outs, expected):                                # expected = the distinct pages a stream is tiled from
        assert o.size % e.size == 0 and np.array_equal(o.reshape(-1, e.size), np.broadcast_to(e, (o.size // e.size, e.size)))
    t0 = time.perf_counter()
    tickets = []
    for b in range(batches):
        tickets.append(st.submit(streams))
        if b >= slots - 1:
            st.wait(tickets[b - (slots - 1)])
    for tk in tickets[-(slots - 1):] if slots > 1 else []:
        st.wait(tk)
    dt = time.perf_counter() - t0
    res[f"slots_{slots}"] = {"seconds": round(dt, 4), "decoded_GBps": round(batches * u_bytes / dt / 1e9, 2),
                             "compressed_in_GBps": round(batches * c_bytes / dt / 1e9, 2)}
    st.close()
print(json.dumps(res))
