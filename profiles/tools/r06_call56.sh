#!/bin/bash
# Round 6, final build: the launch's tail (wave_times.py on the product kernel built with -DBROTLIG_WAVE_TIMES=1) and the benchmark line with this kernel's traffic attached.
export TMPDIR=/tmp
out=gpurun_out/r06_final; mkdir -p $out
for w in mixed text; do BROTLIG_HIP_SO=$(pwd)/build/abv/lib_wavetimes.so python profiles/tools/wave_times.py --workload $w 2>> $out/bench.err; done > $out/wave_times.jsonl
cat $out/wave_times.jsonl | cut -c1-400
timeout 900 python bench.py > $out/bench_with_traffic.json 2>> $out/bench.err; cut -c1-200 $out/bench_with_traffic.json
timeout 600 python bench.py --two-in-flight --no-cpu-baseline --no-alt-parse 2>> $out/bench.err > $out/bench_two_in_flight.json; grep -o '"two_batches_in_flight": {[^}]*}' $out/bench_two_in_flight.json
