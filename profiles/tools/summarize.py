#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by collect.sh on the MI355X box) into the committed summaries
profiles/r<NN>_<tag>_*.  Usage: python profiles/tools/summarize.py <tag> <round> "<kernel label>"."""
import csv, glob, json, os, shutil, sys

tag, rnd, label = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
pre = os.path.join(dst, f"r{rnd}_{tag}_")

for name in ("bench", "bench_bc3", "bench_runs", "bench_text", "bench_samples16", "bench_records", "bench_distinct4096", "bench_files", "bench_files_optimal_parse", "latency", "streamer_bench", "cpu_decode", "config5_projection"):
    p = os.path.join(src, name + ".json")
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, pre + name + ".json")
shutil.copy(os.path.join(src, "phase_profile.jsonl"), pre + "phase_profile.jsonl")
for extra in ("page_latency.jsonl", "wave_times.jsonl", "sched_times.jsonl", "many_streams.jsonl", "many_textures.jsonl"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), pre + extra)
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])

stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
with open(pre + "kernel_trace_stats.md", "w") as f:
    f.write(f"# Round {int(rnd)}, {label} -- rocprofv3 --kernel-trace --stats\n\n")
    f.write("Command (MI355X box, from /tmp with TMPDIR=/tmp): `rocprofv3 --kernel-trace --stats --output-format csv "
            "-d gpurun_out/%s/trace -o f -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline`\n" % tag)
    f.write("(default workload: %s).  Source: `f_kernel_stats.csv`.\n\n" % bench["config"]["workload"])
    f.write("| kernel | calls | total ns | average ns | % | min ns | max ns |\n|---|---|---|---|---|---|---|\n")
    dec_avg = None
    for r in rows[:10]:
        f.write("| `%s` | %s | %s | %d | %s | %s | %s |\n" % (r["Name"][:80], r["Calls"], r["TotalDurationNs"],
                float(r["AverageNs"]), r["Percentage"], r["MinNs"], r["MaxNs"]))
        if "brotlig_decode_kernel" in r["Name"] and dec_avg is None:
            dec_avg = float(r["AverageNs"]) / 1e6
    f.write("\n`brotlig_decode_kernel` average under rocprofv3: %.3f ms (2 warm-up + 5 timed + 1 verification launch).  "
            "bench.py's own HIP-event average over the 5 timed launches of the un-profiled run of the same "
            "build: %.3f ms (`%s`).\n" % (dec_avg, bench["roofline"]["kernel_ms"], os.path.basename(pre + "bench.json")))

bc3 = glob.glob(os.path.join(src, "trace_bc3", "**", "*kernel_stats.csv"), recursive=True)
if bc3:
    rows3 = list(csv.DictReader(open(bc3[0])))
    with open(pre + "kernel_trace_stats_bc3.md", "w") as f:
        f.write(f"# Round {int(rnd)}, {label} -- rocprofv3 --kernel-trace --stats, config 4 (256 BC3 textures, 4 GiB)\n\n")
        f.write("Command (MI355X box, from /tmp with TMPDIR=/tmp): `rocprofv3 --kernel-trace --stats --output-format csv "
                "-d gpurun_out/%s/trace_bc3 -o f -- python bench.py --workload bc3 --streams 256 --steps 5 --warmup 2 --no-cpu-baseline`\n\n" % tag)
        f.write("| kernel | calls | total ns | average ns | % | min ns | max ns |\n|---|---|---|---|---|---|---|\n")
        for r in rows3[:6]:
            f.write("| `%s` | %s | %s | %d | %s | %s | %s |\n" % (r["Name"][:80], r["Calls"], r["TotalDurationNs"],
                    float(r["AverageNs"]), r["Percentage"], r["MinNs"], r["MaxNs"]))
        dk = [float(r["AverageNs"]) for r in rows3 if "brotlig_decondition_kernel" in r["Name"]]
        if dk:
            f.write("\n`brotlig_decondition_kernel`: %.3f ms for 4 GiB read + 4 GiB written = %.2f TB/s.\n" % (dk[0] / 1e6, 2 * 4294967296.0 / dk[0] / 1e3))

def pmc(sub, counter):
    p = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(p))
            if r["Counter_Name"] == counter and "brotlig_decode_kernel" in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)

fetch, nf = pmc("pmc_fetch", "FETCH_SIZE")
write, nw = pmc("pmc_write", "WRITE_SIZE")
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
j = {
    "kernel": "brotlig_decode_kernel", "build": label, "workload": bench["config"]["workload"],
    "kernel_source_sha16": bench["roofline"].get("kernel_source_sha16"),
    "kernel_disasm_sha16": bench["roofline"].get("kernel_disasm_sha16"),     # what bench.py gates the reuse of this measurement on
    "launches_averaged": {"FETCH_SIZE": nf, "WRITE_SIZE": nw},
    "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
    "fetch_bytes_per_launch": fetch * 1024, "write_bytes_per_launch": write * 1024,
    "traffic_bytes_per_launch_raw": (fetch + write) * 1024,
    "traffic_bytes_per_launch_fetch_doubled": (2 * fetch + write) * 1024,
    "traffic_bytes_per_launch_calibrated": (2 * fetch + write) * 1024,
    "algorithmic_bytes_per_launch": alg,
    "note": "two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only. "
            "Calibrated in round 3 on known byte counts in this kernel's own access patterns (profiles/r03_traffic_calibration.md, "
            "profiles/tools/traffic_calib.hip): every read request beyond L2 is a 128-byte line that FETCH_SIZE tallies as 64 bytes "
            "(8-byte per-lane loads as well as wide ones), WRITE_SIZE counts 64-byte requests correctly: "
            "traffic = 2 x FETCH_SIZE + WRITE_SIZE.",
}
try:                    # the requests behind the two derived counters, and the L2 hit rate (one more pass)
    rd, _ = pmc("pmc_tcc", "TCC_EA0_RDREQ_sum"); wr, _ = pmc("pmc_tcc", "TCC_EA0_WRREQ_sum")
    hit, _ = pmc("pmc_tcc", "TCC_HIT_sum"); miss, _ = pmc("pmc_tcc", "TCC_MISS_sum")
    j["tcc"] = {"TCC_EA0_RDREQ_sum": rd, "TCC_EA0_WRREQ_sum": wr, "TCC_HIT_sum": hit, "TCC_MISS_sum": miss,
                "read_bytes_beyond_l2": rd * 128, "write_bytes_beyond_l2": wr * 64, "l2_hit_rate": hit / (hit + miss) if hit + miss else None}
except Exception as e:
    j["tcc"] = {"error": str(e)[:80]}
json.dump(j, open(pre + "hbm_traffic.json", "w"), indent=1)
# SQ counters (two passes), decode kernel only, averaged over its dispatches
sq = {}
for sub in ("pmc_sq1", "pmc_sq2", "pmc_sq3"):
    ps = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    if not ps: continue
    acc = {}
    for r in csv.DictReader(open(ps[0])):
        if "brotlig_decode_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items(): sq[k] = sum(v) / len(v)
if sq:
    with open(pre + "pmc_sq_counters.md", "w") as f:
        f.write(f"# Round {int(rnd)}, {label} -- rocprofv3 --pmc SQ counters (two separate passes, kernel-trace only)\n\n")
        f.write("Workload: bench.py default (%s).  Values per dispatch of `brotlig_decode_kernel`, summed over all "
                "SEs/XCDs as rocprofv3 reports them.\n\n| counter | per dispatch |\n|---|---|\n" % bench["config"]["workload"])
        for k in sorted(sq): f.write("| %s | %.4g |\n" % (k, sq[k]))
        if "SQ_WAVE_CYCLES" in sq and "SQ_ACTIVE_INST_VALU" in sq:
            f.write("\nVALU-active / wave cycles: %.3f; wait-any / wave cycles: %.3f; VALU instructions per decompressed GiB: %.3g\n" % (
                sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"], sq.get("SQ_WAIT_ANY", 0) / sq["SQ_WAVE_CYCLES"], sq.get("SQ_INSTS_VALU", 0) / 4.0))
        if "GRBM_GUI_ACTIVE" in sq and "SQ_ACTIVE_INST_VALU" in sq:
            # rocprof's derived VALUBusy / scalar / LDS equivalents: quad-cycles of the arbiter per CU-cycle (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
            cu_cycles = sq["GRBM_GUI_ACTIVE"] / 8.0 * 256.0
            f.write("\nPipe occupancy (quad-cycles / (cycles x 256 CUs)): vector ALU %.3f, scalar %.3f, LDS %.3f; active lanes per vector instruction %.3f\n" % (
                sq["SQ_ACTIVE_INST_VALU"] / cu_cycles, sq.get("SQ_ACTIVE_INST_SCA", 0) / cu_cycles, sq.get("SQ_ACTIVE_INST_LDS", 0) / cu_cycles,
                sq.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * sq["SQ_ACTIVE_INST_VALU"])))
# round 6: instruction cache of the decode kernel (70 KB of code against 64 KB of cache), and the L2 counters with every page distinct
extra = {}
for sub in ("pmc_icache", "pmc_tcc_distinct4096"):
    ps = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    if not ps: continue
    acc = {}
    for r in csv.DictReader(open(ps[0])):
        if "brotlig_decode_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    extra[sub] = {k: sum(v) / len(v) for k, v in acc.items()}
if extra:
    if "pmc_icache" in extra and extra["pmc_icache"].get("SQC_ICACHE_REQ"):
        ic = extra["pmc_icache"]; ic["hit_rate"] = ic.get("SQC_ICACHE_HITS", 0) / ic["SQC_ICACHE_REQ"]
    json.dump(extra, open(pre + "pmc_icache_and_tcc_distinct.json", "w"), indent=1)
print(json.dumps({"decode_ms_rocprof": dec_avg, "decode_ms_bench": bench["roofline"]["kernel_ms"],
                  "fetch_GB": fetch * 1024 / 1e9, "write_GB": write * 1024 / 1e9, "alg_GB": alg / 1e9}))
