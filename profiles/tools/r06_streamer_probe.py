#!/usr/bin/env python3
"""Round 6 probe: where a batch of the device-output streamer spends its host time (submit / consumer enqueue / wait), copy and in-place."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from brotli_g_sdk_amd import api
spb, pps, batches = 4, 1024, 48
streams, expected = bench.build_streams("mixed", range(spb), pps, 128)
c_bytes = sum(len(s) for s in streams); u_bytes = sum(api.DecompressedSize(s) for s in streams)
offs, pos = [], 0
for s_ in streams:
    offs.append(pos); pos = (pos + len(s_) + 15) // 16 * 16
szs = [len(s_) for s_ in streams]
out = {}
for mode in ("copy", "in_place", "in_place_no_consumer"):
    st = api.Streamer(slots=3, slot_in_bytes=c_bytes + (1 << 20), slot_out_bytes=u_bytes + (1 << 20), max_streams=spb, device_output=True)
    consumer = torch.cuda.Stream()
    for _ in range(3):
        area = st.acquire()
        for o, s_ in zip(offs, streams):
            area[o:o + len(s_)] = s_
        tk = st.submit_in_place(offs, szs)
    st.wait(tk); torch.cuda.synchronize()
    t_sub = t_con = 0.0
    t0 = time.perf_counter(); tickets = []
    for b in range(batches):
        a0 = time.perf_counter()
        if mode == "copy":
            tk = st.submit(streams)
        else:
            st.acquire(); tk = st.submit_in_place(offs, szs)
        a1 = time.perf_counter()
        tickets.append(tk)
        if mode != "in_place_no_consumer":
            with torch.cuda.stream(consumer):
                p, sz, ev = st.device_output(tk, 0)
                t = st.device_tensor(tk, 0)
                t[:1 << 20].view(torch.int64).sum()
                st.consumer_done(tk)
        a2 = time.perf_counter()
        t_sub += a1 - a0; t_con += a2 - a1
    for tk in tickets[-3:]:
        st.wait(tk)
    consumer.synchronize()
    dt = time.perf_counter() - t0
    out[mode] = {"decoded_GBps": round(batches * u_bytes / dt / 1e9, 1), "compressed_GBps": round(batches * c_bytes / dt / 1e9, 1), "ms_per_batch": round(dt / batches * 1e3, 3),
                 "submit_ms": round(t_sub / batches * 1e3, 3), "consumer_enqueue_ms": round(t_con / batches * 1e3, 3)}
    st.close()
# raw H2D rate of the same bytes from pinned memory, and the decode alone
h = torch.empty(c_bytes, dtype=torch.uint8).pin_memory(); d = torch.empty(c_bytes, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); out["raw_h2d_GBps"] = round(20 * c_bytes / (time.perf_counter() - t0) / 1e9, 1)
dec = api.BatchDecoder(streams); dec.decode(); tot, k = dec.timed(2, 10)
out["decode_step_ms"] = round(tot / 10, 3); out["batch"] = f"{u_bytes >> 20} MiB decoded, {c_bytes >> 20} MiB compressed"
print(json.dumps(out))
