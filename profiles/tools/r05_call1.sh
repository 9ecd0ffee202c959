#!/bin/bash
# Round 5, first call: the widened GPU parity matrix, the driver-style bench line, the de-conditioning A/B (wide path / gather / round 4's
# kernel), the launch's head and tail (wave times), and a kernel trace of config 4.
out=gpurun_out/r05c1; mkdir -p $out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout 200 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-400 $out/bench.json
timeout 400 python profiles/tools/ab_run.py --workloads bc3 mixed --reps 3 --steps 5 --out $out/ab.json 2> $out/ab.err | tee $out/ab.txt
timeout 120 python profiles/tools/wave_times.py --workload mixed > $out/wave_times_mixed.json 2> $out/wave_times.err; cat $out/wave_times_mixed.json
timeout 120 python profiles/tools/wave_times.py --workload text > $out/wave_times_text.json 2>> $out/wave_times.err; cat $out/wave_times_text.json
root=$(pwd); cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace_bc3 -o t -- python $root/bench.py --workload bc3 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-parse > $root/$out/trace_bc3.log 2>&1
cd $root
find $out -name '*_kernel_trace.csv' -size +4M -delete; find $out -name '*agent_info*' -delete
cat $out/trace_bc3/*/*kernel_stats.csv 2>/dev/null | head -12
