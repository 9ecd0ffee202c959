#!/bin/bash
# PMC passes for the ISA stage budget: the product kernel and its ablation builds (profiles/tools/ablate.sh semantics).
bash profiles/tools/pmc_variants.sh r04budget base abl1 abl2 abl4 abl8 abl16 abl32 abl64 abl128 abl255 2>&1 | tail -14
python - <<'PY'
import csv, glob, json
out = {}
for d in sorted(glob.glob("gpurun_out/r04budget/*_1")):
    v = d.split("/")[-1][:-2]
    for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(p)) if "brotlig_decode_kernel" in r["Kernel_Name"]]
        if rows:
            out[v] = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / len(rows) / 1e6
json.dump(out, open("gpurun_out/r04budget/kernel_ms.json", "w"), indent=1)
print("kernel ms", out)
PY
for k in "mixed 16" "text 16"; do python profiles/phase_profile.py $k; done > gpurun_out/r04budget/phase_profile.jsonl 2>/dev/null; cat gpurun_out/r04budget/phase_profile.jsonl | cut -c1-400
