#!/bin/bash
# after the last kernel change (the two-wavefront kernel's register budget): GPU suite, small-batch latencies, config-4 kernel trace, bench line
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/r05_final; mkdir -p $out
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $out/pytest.log | tail -2
python profiles/tools/latency.py auto one_wavefront two_wavefronts > "$out/latency.json" 2>> "$out/bench.err"
for m in one_wavefront two_wavefronts; do python profiles/tools/page_latency.py $m 2>> "$out/bench.err" | grep "^{"; done > "$out/page_latency.jsonl"
python bench.py > "$out/bench_after.json" 2>> "$out/bench.err"; tail -1 $out/bench_after.json | cut -c1-200
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_bc3" -o f -- python "$root/bench.py" --workload bc3 --streams 256 --steps 5 --warmup 2 --no-cpu-baseline > "$out/trace_bc3.log" 2>&1
find "$out" -name '*_kernel_trace.csv' -size +8M -delete; find "$out" -name '*agent_info*' -delete
