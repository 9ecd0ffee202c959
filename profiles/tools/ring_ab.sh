#!/bin/bash
# A/B of the LDS-staged sub-stream ring (RingReader) against the register bit reader (BitReader) in the entropy kernel of the
# split experiment: the same kernel, the same LDS footprint (control build padded by the ring's 4 KiB), kernel times from rocprofv3.
set -u
tag=${1:-ring}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cs=$root/brotli_g_sdk_amd/csrc
build() { local name=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DBROTLIG_WITH_SPLIT -I "$root/profiles/experiments/split_path" -I "$root/include" -I "$cs" "$@" -o "$out/lib_$name.so" "$cs/brotlig_hip.hip" "$cs/brotlig_streamer.hip" 2>> "$out/build.err"; }
build regs
build regs_padded -DBROTLIG_E_PAD_LDS
build ring -DBROTLIG_E_RING=1
cd /tmp
for w in mixed text; do for v in regs regs_padded ring; do
  BROTLIG_SPLIT=1 BROTLIG_HIP_SO="$out/lib_$v.so" rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_${v}_$w" -o f -- python "$root/bench.py" --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-alt-parse > "$out/trace_${v}_$w.log" 2>&1
done; done
find "$out" -name '*_kernel_trace.csv' -delete; find "$out" -name '*agent_info*' -delete; rm -f "$out"/lib_*.so
cd "$root"
python - <<PY
import glob, os, csv, json
out = "$out"
for w in ("mixed", "text"):
    for v in ("regs", "regs_padded", "ring"):
        ok = None
        for line in open(os.path.join(out, f"trace_{v}_{w}.log"), errors="ignore"):
            if line.startswith("{"): ok = json.loads(line)["bit_exact"]
        for p in glob.glob(os.path.join(out, f"trace_{v}_{w}", "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(p)):
                if "entropy" in r["Name"]: print(w, v, "entropy kernel avg ms", round(float(r["AverageNs"]) / 1e6, 3), "calls", r["Calls"], "bit_exact", ok)
PY
tail -n 3 "$out/build.err"
