#!/bin/bash
# Second A/B of the split path: variants of the assembly kernel (loads issued a step ahead, far loads ablated, longer history).
#   bash profiles/tools/split_ab2.sh <tag>
set -u
tag=${1:-split2}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cs=$root/brotli_g_sdk_amd/csrc
build() { # name flags...
  local name=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DBROTLIG_WITH_SPLIT -I "$root/profiles/experiments/split_path" -I "$root/include" -I "$cs" "$@" -o "$out/lib_$name.so" "$cs/brotlig_hip.hip" "$cs/brotlig_streamer.hip" 2>> "$out/build.err"
}
build p2 -DBROTLIG_E_WAVES=5 -DBROTLIG_L_WAVES=5 -DBROTLIG_L_PREFETCH=2
build p1 -DBROTLIG_E_WAVES=5 -DBROTLIG_L_WAVES=5 -DBROTLIG_L_PREFETCH=1
build p0 -DBROTLIG_E_WAVES=5 -DBROTLIG_L_WAVES=5 -DBROTLIG_L_PREFETCH=0
build p2l6 -DBROTLIG_E_WAVES=5 -DBROTLIG_L_WAVES=6 -DBROTLIG_L_PREFETCH=2
build p2l4 -DBROTLIG_E_WAVES=5 -DBROTLIG_L_WAVES=4 -DBROTLIG_L_PREFETCH=2
build p2nofar -DBROTLIG_E_WAVES=5 -DBROTLIG_L_WAVES=5 -DBROTLIG_L_PREFETCH=2 -DBROTLIG_ABLATE=8
build p2hist -DBROTLIG_E_WAVES=5 -DBROTLIG_L_WAVES=5 -DBROTLIG_L_PREFETCH=2 -DBROTLIG_TUNE_SHORT_COPY=32 -DBROTLIG_TUNE_OWN_COPY=128 -DBROTLIG_TUNE_HIST=1040 -DBROTLIG_TUNE_ROUND_MAX=512 -DBROTLIG_TUNE_WIN=2000 -DBROTLIG_TUNE_DIST_LUT_BITS=8
for w in mixed text; do
  for v in p2 p1 p0 p2l6 p2l4 p2nofar p2hist; do
    BROTLIG_SPLIT=1 BROTLIG_HIP_SO="$out/lib_$v.so" python bench.py --workload $w --no-cpu-baseline --no-alt-parse > "$out/split_${v}_$w.json" 2>> "$out/bench.err"
  done
done
# entropy kernel occupancy sweep (grid per CU) and assembly sweep on the p2 build
for e in 12 16 20; do BROTLIG_E_PER_CU=$e BROTLIG_SPLIT=1 BROTLIG_HIP_SO="$out/lib_p2.so" python bench.py --no-cpu-baseline --no-alt-parse > "$out/sweepE_${e}_mixed.json" 2>> "$out/bench.err"; done
for l in 12 16 20; do BROTLIG_L_PER_CU=$l BROTLIG_SPLIT=1 BROTLIG_HIP_SO="$out/lib_p2.so" python bench.py --no-cpu-baseline --no-alt-parse > "$out/sweepL_${l}_mixed.json" 2>> "$out/bench.err"; done
cd /tmp
BROTLIG_SPLIT=1 BROTLIG_HIP_SO="$out/lib_p2.so" rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_p2" -o f -- python "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-alt-parse > "$out/trace_p2.log" 2>&1
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
for p in glob.glob(os.path.join(out, "trace_p2", "**", "*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(p)))[:3]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
tail -3 "$out/bench.err" "$out/build.err"
