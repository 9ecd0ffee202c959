#!/usr/bin/env python3
"""How much on-chip history would the LZ77 assembly need?  Runs the entropy kernel of the split path on the CPU simulator over
pages of the benchmark's data classes (tests/sim; no GPU needed), takes the command arrays and reports, per class:
  * commands per page, copies per page, mean copy length, literal share
  * the share of copies (and of copy bytes) whose distance exceeds H for a range of H
  * 128-byte source lines fetched per page when every copy further back than H bytes fetches its source lines from memory
    (the fused kernel's situation with H ~ 528..1000: no reuse on chip), and when an LRU cache of K lines per page sits in between.
Usage: python profiles/tools/far_reuse.py [pages_per_class=24] [flags=0]"""
import ctypes, json, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from brotli_g_sdk_amd import datagen as D, encoder as E   # noqa: E402
import test_sim_split as T                               # noqa: E402

PAGES = int(sys.argv[1]) if len(sys.argv) > 1 else 24
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sim = T.build_sim("libbrotlig_sim.so")
sim.sim_entropy_batch.restype = ctypes.c_int
sim.sim_entropy_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
HS = [528, 1024, 2048, 4096, 8192, 16384, 32768]
KS = [0, 16, 32, 64, 128, 256, 512]
M = (1 << 18) - 1


def analyse(kind):
    data = {"mixed": D.mixed, "text": D.text, "records": D.records, "samples16": D.samples16, "runs": D.runs}[kind](PAGES * 65536, 0)
    stream = E.encode(data, flags=FLAGS)
    cmds, lits, hdr, *_ = T.run_entropy(sim, [stream], [len(data)])
    ncmd = ncopy = copy_bytes = lit_bytes = pages = 0
    far_c = np.zeros(len(HS)); far_b = np.zeros(len(HS))
    lines = np.zeros((len(HS), len(KS)))
    for g in range(PAGES):
        n, flags = int(hdr[g, 0]), int(hdr[g, 1])
        if not flags & 1:
            continue
        pages += 1
        c = cmds[g, :n + 1].astype(np.int64)
        o, lp, d = c & M, (c >> 18) & M, (c >> 36) & M
        ins = lp[1:] - lp[:-1]; tot = o[1:] - o[:-1]; cp = tot - ins
        dist = d[:-1]; dst = o[:-1] + ins
        has = cp > 0
        ncmd += n; ncopy += int(has.sum()); copy_bytes += int(cp.sum()); lit_bytes += int(ins.sum())
        for hi, H in enumerate(HS):
            far = has & (dist > H)
            far_c[hi] += far.sum(); far_b[hi] += cp[far].sum()
            # source lines of the far copies, in command order
            src0 = (dst - dist)[far]; ln = np.minimum(cp, dist)[far]
            seq = []
            for s0, l in zip(src0.tolist(), ln.tolist()):
                seq.extend(range(s0 >> 7, ((s0 + l - 1) >> 7) + 1))
            for ki, K in enumerate(KS):
                if K == 0:
                    lines[hi, ki] += len(seq); continue
                lru, miss = {}, 0
                for t, x in enumerate(seq):
                    if x in lru:
                        lru[x] = t
                    else:
                        miss += 1
                        if len(lru) >= K:
                            del lru[min(lru, key=lru.get)]
                        lru[x] = t
                lines[hi, ki] += miss
    p = max(pages, 1)
    return {"class": kind, "pages": pages, "ratio": round(len(data) / len(stream), 2), "commands_per_page": round(ncmd / p),
            "copies_per_page": round(ncopy / p), "mean_copy_len": round(copy_bytes / max(ncopy, 1), 1), "literal_share": round(lit_bytes / (lit_bytes + copy_bytes), 3),
            "share_of_copies_further_than": {str(H): round(far_c[i] / max(ncopy, 1), 3) for i, H in enumerate(HS)},
            "share_of_copy_bytes_further_than": {str(H): round(far_b[i] / max(copy_bytes, 1), 3) for i, H in enumerate(HS)},
            "line_fetches_per_page": {f"history {H}": {("no cache" if K == 0 else f"LRU {K} lines"): round(lines[i, k] / p) for k, K in enumerate(KS)} for i, H in enumerate(HS[:4])}}


for kind in (sys.argv[3:] or ["mixed", "text", "records", "samples16"]):
    print(json.dumps(analyse(kind)))
