#!/usr/bin/env python3
"""Folds rocprofv3 PC-sampling CSV output (one row per sampled wavefront) into a histogram per code-object offset, small
enough to travel back from the GPU box.  usage: pcs_aggregate.py <dir with *pc_sampling*.csv> <out.json>"""
import collections, csv, glob, json, os, sys
src, out = sys.argv[1], sys.argv[2]
res = {"files": [], "hist": {}}
for path in glob.glob(os.path.join(src, "**", "*pc_sampling*.csv"), recursive=True):
    with open(path, newline="") as f:
        rd = csv.reader(f)
        hdr = next(rd, None)
        if not hdr:
            continue
        res["files"].append({"path": path, "header": hdr})
        low = [h.lower() for h in hdr]
        keycols = [i for i, h in enumerate(low) if any(k in h for k in ("offset", "instruction", "stall", "inst_type", "reason", "issued", "dual"))]
        if not keycols:
            keycols = list(range(len(hdr)))
        cnt = collections.Counter()
        n = 0
        head = []
        for row in rd:
            if n < 40:
                head.append(row)
            n += 1
            cnt[tuple(row[i] for i in keycols if i < len(row))] += 1
        res["hist"][os.path.basename(path)] = {"rows": n, "key_columns": [hdr[i] for i in keycols], "sample_rows": head,
                                               "counts": [[list(k), v] for k, v in cnt.most_common(20000)]}
json.dump(res, open(out, "w"))
print("pc sampling:", {k: v["rows"] for k, v in res["hist"].items()})
