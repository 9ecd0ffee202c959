#!/bin/bash
# Fourth A/B of the split path: the page-in-LDS assembly kernel (one workgroup per page; BROTLIG_SPLIT=3).
set -u
tag=${1:-split4}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cs=$root/brotli_g_sdk_amd/csrc
build() { local name=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DBROTLIG_WITH_SPLIT -I "$root/profiles/experiments/split_path" -I "$root/include" -I "$cs" "$@" -o "$out/lib_$name.so" "$cs/brotlig_hip.hip" "$cs/brotlig_streamer.hip" 2>> "$out/build.err"; }
build p4 -DBROTLIG_PAGE_WAVES=4
build p8 -DBROTLIG_PAGE_WAVES=8
build p16 -DBROTLIG_PAGE_WAVES=16
( export BROTLIG_SPLIT=3 BROTLIG_HIP_SO="$out/lib_p4.so"; python -m pytest tests/test_gpu_decode.py -m gpu -q -k "not 128k and not many_command_shapes" ) > "$out/pytest_split3.log" 2>&1
tail -6 "$out/pytest_split3.log"
for w in mixed text records runs samples16; do
  python bench.py --workload $w --no-cpu-baseline --no-alt-parse > "$out/fused_$w.json" 2>> "$out/bench.err"
  for v in p4 p8 p16; do
    BROTLIG_SPLIT=3 BROTLIG_HIP_SO="$out/lib_$v.so" python bench.py --workload $w --no-cpu-baseline --no-alt-parse > "$out/split_${v}_$w.json" 2>> "$out/bench.err"
  done
done
cd /tmp
for v in p4 p8; do
BROTLIG_SPLIT=3 BROTLIG_HIP_SO="$out/lib_$v.so" rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$v" -o f -- python "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-alt-parse > "$out/trace_$v.log" 2>&1
done
BROTLIG_SPLIT=3 BROTLIG_HIP_SO="$out/lib_p8.so" rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_p8" -o f -- python "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-alt-parse > "$out/pmc_p8.log" 2>&1
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
for t in ("trace_p4", "trace_p8"):
    for p in glob.glob(os.path.join(out, t, "**", "*kernel_stats.csv"), recursive=True):
        for r in list(csv.DictReader(open(p)))[:3]:
            print(t, r["Name"][:60], r["Calls"], r["AverageNs"])
for p in glob.glob(os.path.join(out, "pmc_p8", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for r in csv.DictReader(open(p)):
        if "assemble" in r["Kernel_Name"] or "entropy" in r["Kernel_Name"]:
            acc.setdefault((r["Kernel_Name"][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()): print(k, sum(v) / len(v))
PY
tail -n 3 "$out/bench.err" "$out/build.err"
