#!/bin/bash
# Fused kernel vs the split path (entropy kernel -> command/literal arrays -> assembly kernel), VERDICT r2 item 3.
#   bash profiles/tools/split_ab.sh <tag>        (MI355X box, from the repo root, through gpurun)
# Builds variants of the library with different register budgets for the two kernels, runs the parity tests on the split
# path, then bench.py per workload for the fused kernel and every variant, and a rocprofv3 kernel trace of the best-looking one.
set -u
tag=${1:-split}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cs=$root/brotli_g_sdk_amd/csrc
build() { # name E L
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DBROTLIG_WITH_SPLIT -I "$root/profiles/experiments/split_path" -I "$root/include" -I "$cs" -DBROTLIG_E_WAVES=$2 -DBROTLIG_L_WAVES=$3 \
        -o "$out/lib_$1.so" "$cs/brotlig_hip.hip" "$cs/brotlig_streamer.hip" 2>> "$out/build.err"
}
build e5l5 5 5; build e5l6 5 6; build e5l8 5 8; build e4l6 4 6
# parity on the split path (default build = e5l6): the C-ABI tests with the split path switched on
( export BROTLIG_SPLIT=1 BROTLIG_HIP_SO="$out/lib_e5l6.so"; python -m pytest tests/test_gpu_decode.py tests/test_gpu_differential.py -m gpu -x -q ) > "$out/pytest_split.log" 2>&1
tail -3 "$out/pytest_split.log"
for w in mixed text records runs; do
  s=16; [ $w = runs ] && s=16
  python bench.py --workload $w --streams $s --no-cpu-baseline --no-alt-parse > "$out/fused_$w.json" 2>> "$out/bench.err"
  for v in e5l5 e5l6 e5l8 e4l6; do
    BROTLIG_SPLIT=1 BROTLIG_HIP_SO="$out/lib_$v.so" python bench.py --workload $w --streams $s --no-cpu-baseline --no-alt-parse > "$out/split_${v}_$w.json" 2>> "$out/bench.err"
  done
done
cd /tmp
BROTLIG_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_split" -o f -- python "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-alt-parse > "$out/trace_split.log" 2>&1
BROTLIG_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_split_text" -o f -- python "$root/bench.py" --workload text --steps 5 --warmup 2 --no-cpu-baseline --no-alt-parse > "$out/trace_split_text.log" 2>&1
find "$out" -name '*_kernel_trace.csv' -size +4M -delete; find "$out" -name '*agent_info*' -delete; rm -f "$out"/lib_*.so
cd "$root"
python - <<PY
import json, glob, os
out = "$out"
for w in ("mixed", "text", "records", "runs"):
    row = {}
    for f in sorted(glob.glob(os.path.join(out, f"*_{w}.json"))):
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
            row[os.path.basename(f)[:-len(w) - 6]] = (j["value"], j["roofline"]["kernel_ms"], j["bit_exact"])
        except Exception as e:
            row[os.path.basename(f)] = str(e)[:60]
    print(w, row)
import csv
for t in ("trace_split", "trace_split_text"):
    for p in glob.glob(os.path.join(out, t, "**", "*kernel_stats.csv"), recursive=True):
        for r in list(csv.DictReader(open(p)))[:4]:
            print(t, r["Name"][:60], r["Calls"], r["AverageNs"])
PY
tail -5 "$out/bench.err" "$out/build.err"
