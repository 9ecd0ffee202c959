#!/usr/bin/env python3
"""gpurun_out/<tag>/ (profiles/tools/traffic_calib.sh) -> one JSON: counter values per launch for every calibration kernel and
for brotlig_decode_kernel, and the derived bytes-per-known-byte factors.  Usage: traffic_calib_summary.py <tag>"""
import csv, glob, json, os, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "gpurun_out", tag)


def counters(prefix, kernel_substr):
    acc = {}
    for d in sorted(glob.glob(os.path.join(src, prefix + "_set*"))):
        if not os.path.isdir(d):
            continue
        for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(p)):
                if kernel_substr in r["Kernel_Name"]:
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


out = {}
for k, sub in (("wide16", "wide16"), ("sub8", "sub8"), ("flush16", "flush_far"), ("far", "flush_far"), ("farx", "flush_far")):
    p = os.path.join(src, k + "_plain.json")
    if not os.path.exists(p) or not os.path.getsize(p):
        continue
    known = json.loads(open(p).read().strip().splitlines()[-1])
    c = counters(k, sub)
    row = {"known": known, "counters_per_launch": c}
    if "FETCH_SIZE" in c:
        row["fetch_bytes"] = c["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in c:
        row["write_bytes"] = c["WRITE_SIZE"] * 1024
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        row["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    out[k] = row
dec = counters("decode", "brotlig_decode_kernel")
if dec:
    row = {"counters_per_launch": dec}
    if dec.get("TCC_HIT_sum") is not None and dec.get("TCC_MISS_sum"):
        row["l2_hit_rate"] = dec["TCC_HIT_sum"] / (dec["TCC_HIT_sum"] + dec["TCC_MISS_sum"])
    for n in ("decode_set0",):
        for line in open(os.path.join(src, n + ".log"), errors="ignore"):
            if line.startswith("{") and "roofline" in line:
                b = json.loads(line)
                row["bench"] = {"kernel_ms": b["roofline"]["kernel_ms"], "algorithmic_bytes": b["roofline"]["algorithmic_bytes_per_launch"],
                                "kernel_source_sha16": b["roofline"]["kernel_source_sha16"], "workload": b["config"]["workload"]}
    out["brotlig_decode_kernel"] = row
print(json.dumps(out, indent=1))
