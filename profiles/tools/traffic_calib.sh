#!/bin/bash
# Calibrate FETCH_SIZE / WRITE_SIZE / TCC hit counters on known byte counts in the decode kernel's access patterns, then read the
# decode kernel itself with the same counter sets (MI355X box; run through gpurun from the repo root):
#   bash profiles/tools/traffic_calib.sh <tag>
# PMC passes are separate runs with --kernel-trace only.  profiles/tools/traffic_calib_summary.py writes the committed report.
set -u
tag=${1:-calib}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o "$root/profiles/tools/traffic_calib" "$root/profiles/tools/traffic_calib.hip" || exit 1
sets=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_WRITE_sum")
cd /tmp
for k in wide16 sub8 flush16 far farx; do
  "$root/profiles/tools/traffic_calib" $k 1 3 > "$out/${k}_plain.json" 2>> "$out/calib.err"
  i=0
  for s in "${sets[@]}"; do
    rocprofv3 --kernel-trace --pmc $s --output-format csv -d "$out/${k}_set$i" -o f -- "$root/profiles/tools/traffic_calib" $k 1 2 > "$out/${k}_set$i.log" 2>&1
    i=$((i+1))
  done
done
# the decode kernel, default benchmark workload, same counter sets (FETCH/WRITE come from collect.sh's passes too)
i=0
for s in "${sets[@]}"; do
  rocprofv3 --kernel-trace --pmc $s --output-format csv -d "$out/decode_set$i" -o f -- python "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-alt-parse > "$out/decode_set$i.log" 2>&1
  i=$((i+1))
done
find "$out" -name '*_kernel_trace.csv' -size +4M -delete
find "$out" -name '*agent_info*' -delete
python "$root/profiles/tools/traffic_calib_summary.py" "$tag" > "$out/summary.json" 2>> "$out/calib.err"
tail -5 "$out/calib.err"; cat "$out/summary.json" | head -80
