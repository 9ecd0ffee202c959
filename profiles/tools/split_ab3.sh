#!/bin/bash
# Third A/B of the split path: in-place assembly kernel (no LDS window, one wavefront per page; BROTLIG_SPLIT=2).
set -u
tag=${1:-split3}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cs=$root/brotli_g_sdk_amd/csrc
build() { local name=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DBROTLIG_WITH_SPLIT -I "$root/profiles/experiments/split_path" -I "$root/include" -I "$cs" "$@" -o "$out/lib_$name.so" "$cs/brotlig_hip.hip" "$cs/brotlig_streamer.hip" 2>> "$out/build.err"; }
build g8 -DBROTLIG_G_WAVES=8
build g6 -DBROTLIG_G_WAVES=6
( export BROTLIG_SPLIT=2 BROTLIG_HIP_SO="$out/lib_g8.so"; python -m pytest tests/test_gpu_decode.py tests/test_gpu_differential.py -m gpu -q ) > "$out/pytest_split2.log" 2>&1
tail -8 "$out/pytest_split2.log"
for w in mixed text records runs samples16; do
  python bench.py --workload $w --no-cpu-baseline --no-alt-parse > "$out/fused_$w.json" 2>> "$out/bench.err"
  for v in g8 g6; do
    BROTLIG_SPLIT=2 BROTLIG_HIP_SO="$out/lib_$v.so" python bench.py --workload $w --no-cpu-baseline --no-alt-parse > "$out/split_${v}_$w.json" 2>> "$out/bench.err"
  done
done
for g in 16 24 32; do BROTLIG_G_PER_CU=$g BROTLIG_SPLIT=2 BROTLIG_HIP_SO="$out/lib_g8.so" python bench.py --no-cpu-baseline --no-alt-parse > "$out/sweepG_${g}_mixed.json" 2>> "$out/bench.err"; done
python bench.py --workload bc3 --streams 256 --no-cpu-baseline > "$out/fused_bc3.json" 2>> "$out/bench.err"
BROTLIG_SPLIT=2 BROTLIG_HIP_SO="$out/lib_g8.so" python bench.py --workload bc3 --streams 256 --no-cpu-baseline > "$out/split_g8_bc3.json" 2>> "$out/bench.err"
cd /tmp
BROTLIG_SPLIT=2 BROTLIG_HIP_SO="$out/lib_g8.so" rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_g8" -o f -- python "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-alt-parse > "$out/trace_g8.log" 2>&1
BROTLIG_SPLIT=2 BROTLIG_HIP_SO="$out/lib_g8.so" rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_g8" -o f -- python "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-alt-parse > "$out/pmc_g8.log" 2>&1
find "$out" -name '*_kernel_trace.csv' -size +4M -delete; find "$out" -name '*agent_info*' -delete; rm -f "$out"/lib_*.so
cd "$root"
python - <<PY
import json, glob, os, csv
out = "$out"
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j["value"], j["roofline"]["kernel_ms"], j["bit_exact"])
    except Exception as e:
        print(os.path.basename(f), "ERR", str(e)[:60])
for p in glob.glob(os.path.join(out, "trace_g8", "**", "*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(p)))[:3]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
for p in glob.glob(os.path.join(out, "pmc_g8", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for r in csv.DictReader(open(p)):
        if "brotlig" in r["Kernel_Name"]:
            acc.setdefault((r["Kernel_Name"][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()): print(k, sum(v) / len(v))
PY
tail -n 3 "$out/bench.err" "$out/build.err"
