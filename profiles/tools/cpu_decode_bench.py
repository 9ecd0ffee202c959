#!/usr/bin/env python3
"""DecodeCPU (libbrotlig_cpu.so, the product's CPU entry) on the host cores of the box, on streams of the benchmark's
default workload: decompressed GB/s per worker count.  Not part of bench.py (the GPU path and its benchmark never load
the CPU library); the benchmark's `cpu_baseline` stays the oracle, which restates the reference's cost profile."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from brotli_g_sdk_amd import cpu

streams, expected = bench.build_streams("mixed", range(4), 4096, 256)      # 4 streams x 256 MiB
out = {"workload": "4 streams x 4096 pages x 64 KiB, mixed (bench.py default pages)", "host_cpus": os.cpu_count(), "runs": {}}
import ctypes
L = cpu.lib()
sizes = [int(L.DecompressedSize(s.ctypes.data)) for s in streams]
bufs = [np.zeros(n, np.uint8) for n in sizes]                 # output buffers touched once: page faults are not the decoder's
for workers in (1, 16, 32, 64, 128, 0):
    best = 0.0
    for rep in range(3):
        t0 = time.perf_counter(); total = 0
        for s, b, n in zip(streams, bufs, sizes):
            osz = ctypes.c_uint32(n)
            rc = L.BrotligDecodeCPU(len(s), s.ctypes.data, ctypes.byref(osz), b.ctypes.data, workers)
            assert rc == 0
            total += osz.value
        best = max(best, total / (time.perf_counter() - t0) / 1e9)
    out["runs"][str(workers) if workers else "all"] = round(best, 2)
rc, o = cpu.DecodeCPU(streams[0])
exp = np.tile(expected[0], len(o) // len(expected[0]))
out["bit_exact"] = bool(rc == 0 and np.array_equal(o, exp))
print(json.dumps(out))
