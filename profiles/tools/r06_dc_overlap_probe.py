#!/usr/bin/env python3
"""Does one batch's de-conditioning pass run beside the other batch's page decode when the page kernel leaves room on the compute units?
Config 4 (256 BC3 textures of 16 MiB), two batches in flight on two HIP streams, the page kernel's grid pinned (BrotligDebugSetDecodeGrid:
4096 = 16 wavefronts per compute unit = everything, 3840 = 15, 3584 = 14).  Prints one JSON line per grid.
  BROTLIG_ENABLE_DEBUG_KNOBS=1 python profiles/tools/r06_dc_overlap_probe.py [grids ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("BROTLIG_ENABLE_DEBUG_KNOBS", "1")
import torch
import bench
from brotli_g_sdk_amd import api

grids = [int(x) for x in sys.argv[1:]] or [0, 3840, 3584, 3072]
streams, expected = bench.build_streams("bc3", list(range(256)), 256, 8)
sizes = [len(e) for e in expected]
dev = "cuda:0"
dec_a = api.BatchDecoder(streams, device=dev, out_sizes=sizes)
dec_b = api.BatchDecoder(streams, device=dev, out_sizes=sizes)
s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
u = dec_a.decompressed_bytes
for grid in grids:
    api.DebugSetDecodeGrid(grid)
    # one batch at a time
    for _ in range(2): dec_a.decode(check=False)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(6): dec_a.decode(check=False)
    torch.cuda.synchronize(); one = (time.perf_counter() - t) / 6
    for d_, st_ in ((dec_a, s_a), (dec_b, s_b)):
        with torch.cuda.stream(st_): d_.decode(check=False)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 12
    for k in range(n):
        with torch.cuda.stream(s_a if k % 2 == 0 else s_b): (dec_a if k % 2 == 0 else dec_b).decode(check=False)
    torch.cuda.synchronize(); two = (time.perf_counter() - t) / n
    dec_a.status(); dec_b.status()
    ok = all(bool((dec_b.d_out[dec_b.out_offs[k]:dec_b.out_offs[k] + dec_b.sizes[k]] == torch.from_numpy(expected[k]).to(dev)).all()) for k in range(0, 256, 37))
    print(json.dumps({"decode_grid": grid or "default", "dc_per_cu": os.environ.get("BROTLIG_DC_PER_CU", "default"),
                      "one_batch_ms": round(one * 1e3, 3), "one_batch_GBps": round(u / one / 1e9, 1),
                      "two_in_flight_ms_per_batch": round(two * 1e3, 3), "two_in_flight_GBps": round(u / two / 1e9, 1), "bit_exact_sample": ok}), flush=True)
api.DebugSetDecodeGrid(0)
