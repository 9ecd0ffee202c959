#!/usr/bin/env python3
"""Host-to-host throughput of the streaming front end (BrotligStreamer*): compressed streams in host
memory in, decoded bytes in (pinned) host memory out, PCIe both ways.  Prints one JSON line.
Usage: python profiles/tools/streamer_bench.py [batches=12] [streams_per_batch=4] [pages_per_stream=1024]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from brotli_g_sdk_amd import api

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 48
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
# Round 6: device-output mode -- nothing but the compressed bytes crosses PCIe; a consumer on its own stream reads every batch on the device
# (a checksum kernel over the decoded bytes) and hands the slot back
import torch
for slots in (1, 3):
    st = api.Streamer(slots=slots, slot_in_bytes=c_bytes + (1 << 20), slot_out_bytes=u_bytes + (1 << 20), max_streams=spb, device_output=True)
    consumer = torch.cuda.Stream()
    t = st.submit(streams)
    for i, e in enumerate(expected):                                # warm-up + correctness
        o = st.device_tensor(t, i).cpu().numpy()
        assert o.size % e.size == 0 and np.array_equal(o.reshape(-1, e.size), np.broadcast_to(e, (o.size // e.size, e.size)))
    st.wait(t)
    sums = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tickets = []
    for b in range(batches):
        tk = st.submit(streams)
        tickets.append(tk)
        with torch.cuda.stream(consumer):
            # (one reduction kernel over the batch's first stream -- the consumer is there to be waited for, not to be measured; four tensor
            # constructions and reductions per batch cost more host time than the submit, r06_streamer_probe)
            sums.append(st.device_tensor(tk, 0)[:8 << 20].view(torch.int64).sum())
            st.consumer_done(tk)
    for tk in tickets[-slots:]:
        st.wait(tk)
    consumer.synchronize()
    dt = time.perf_counter() - t0
    assert all(int(x) == int(sums[0]) for x in sums)
    res[f"device_out_slots_{slots}"] = {"seconds": round(dt, 4), "decoded_GBps": round(batches * u_bytes / dt / 1e9, 2),
                                        "compressed_in_GBps": round(batches * c_bytes / dt / 1e9, 2)}
    st.close()
# ... and with the compressed streams already IN the slot's pinned staging area (BrotligStreamerAcquire / SubmitInPlace: where a file read
# leaves them) instead of copied there by Submit -- one host thread, 5 ms per 54 MiB batch, more than upload and decode together
for slots in (3,):
    st = api.Streamer(slots=slots, slot_in_bytes=c_bytes + (1 << 20), slot_out_bytes=u_bytes + (1 << 20), max_streams=spb, device_output=True)
    consumer = torch.cuda.Stream()
    offs, pos = [], 0
    for s_ in streams:
        offs.append(pos); pos = (pos + len(s_) + 15) // 16 * 16
    szs = [len(s_) for s_ in streams]
    for _ in range(slots):                                          # every slot's staging area filled once: the bytes stay where they are
        area = st.acquire()
        for o, s_ in zip(offs, streams):
            area[o:o + len(s_)] = s_
        tk = st.submit_in_place(offs, szs)
    for i, e in enumerate(expected):
        o = st.device_tensor(tk, i).cpu().numpy()
        assert o.size % e.size == 0 and np.array_equal(o.reshape(-1, e.size), np.broadcast_to(e, (o.size // e.size, e.size)))
    st.wait(tk)
    torch.cuda.synchronize()
    sums, tickets = [], []
    t0 = time.perf_counter()
    for b in range(batches):
        st.acquire()                                                # (the "read" has happened: the area holds the streams)
        tk = st.submit_in_place(offs, szs)
        tickets.append(tk)
        with torch.cuda.stream(consumer):
            # (one reduction kernel over the batch's first stream -- the consumer is there to be waited for, not to be measured; four tensor
            # constructions and reductions per batch cost more host time than the submit, r06_streamer_probe)
            sums.append(st.device_tensor(tk, 0)[:8 << 20].view(torch.int64).sum())
            st.consumer_done(tk)
    for tk in tickets[-slots:]:
        st.wait(tk)
    consumer.synchronize()
    dt = time.perf_counter() - t0
    assert all(int(x) == int(sums[0]) for x in sums)
    res[f"device_out_in_place_slots_{slots}"] = {"seconds": round(dt, 4), "decoded_GBps": round(batches * u_bytes / dt / 1e9, 2),
                                                 "compressed_in_GBps": round(batches * c_bytes / dt / 1e9, 2),
                                                 "what": "streams already in the pinned staging area (as a file read leaves them): upload + decode + a consumer kernel per batch"}
    st.close()
print(json.dumps(res))
