#!/usr/bin/env python3
"""Encodes the benchmark's synthetic streams ahead of time (the optimal parse is too slow to run inside
bench.py): python profiles/tools/preencode.py <workload> <streams> <distinct> <flags> <out.npz> [procs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multiprocessing as mp
import numpy as np


def one(job):
    kind, seed, distinct, flags = job
    from brotli_g_sdk_amd import datagen as D, encoder as E
    gen = {"mixed": lambda: D.mixed(distinct * 65536, seed), "text": lambda: D.text(distinct * 65536, seed),
           "records": lambda: D.records(distinct * 65536, seed), "samples16": lambda: D.samples16(distinct * 65536, seed),
           "runs": lambda: D.runs(distinct * 65536, seed + 1)}[kind]
    return seed, E.encode(gen(), flags=flags)


if __name__ == "__main__":
    kind, n, distinct, flags, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    procs = int(sys.argv[6]) if len(sys.argv) > 6 else os.cpu_count()
    with mp.Pool(procs) as pool:
        res = pool.map(one, [(kind, s, distinct, flags) for s in range(n)], chunksize=1)
    np.savez(out, **{str(s): a for s, a in res})
    print("wrote", out, sum(len(a) for _, a in res), "bytes for", n, "streams")
