#!/bin/bash
# After a change that leaves the decode kernel's code as it was but moves the hash of its sources (comments, dead alternatives removed): the bench
# line and the traffic passes of collect.sh again, into the same gpurun_out/<tag>/, so that summarize.py re-issues hbm_traffic.json for the new hash.
#   bash profiles/tools/collect_traffic.sh <tag>
set -u
tag=${1:-final}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
python bench.py > "$out/bench.json" 2> "$out/bench.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o f -- python "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$out/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o f -- python "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o f -- python "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_tcc" -o f -- python "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-alt-parse > "$out/pmc_tcc.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_bc3" -o f -- python "$root/bench.py" --workload bc3 --streams 256 --steps 5 --warmup 2 --no-cpu-baseline > "$out/trace_bc3.log" 2>&1
find "$out" -name '*_kernel_trace.csv' -size +8M -delete; find "$out" -name '*agent_info*' -delete
