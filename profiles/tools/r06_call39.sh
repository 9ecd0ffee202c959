#!/bin/bash
# Round 6, after the no-unroll build: the driver's sequence on the product -- GPU tests, smoke, default bench -- and the benchmark classes.
export TMPDIR=/tmp
out=gpurun_out/r06c39; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $out/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $out/smoke.txt
timeout 900 python bench.py 2>$out/bench.err | tee $out/bench.json | cut -c1-400
for w in text records samples16 runs files; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-alt-parse 2>>$out/bench.err | tee $out/bench_$w.json | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], d['bit_exact'])"
done
timeout 600 python bench.py --workload bc3 --no-cpu-baseline --no-alt-parse 2>>$out/bench.err | tee $out/bench_bc3.json | cut -c1-200
tail -3 $out/bench.err
