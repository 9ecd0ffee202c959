#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06c8; mkdir -p $out; root=$(pwd)
( timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "streamer" ) > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout 300 python profiles/tools/streamer_bench.py 24 > $out/streamer_bench.json 2>>$out/err.log; cat $out/streamer_bench.json
for pol in 1 2 4; do BROTLIG_POLICY=$pol timeout 400 python bench.py --workload files --no-cpu-baseline --steps 5 --warmup 2 2>>$out/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('files policy $pol value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'])
"; done | tee $out/files_policy.txt
for pol in 1 4; do BROTLIG_POLICY=$pol timeout 400 python bench.py --no-alt-parse --no-cpu-baseline --steps 5 --warmup 2 2>>$out/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('mixed policy $pol value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'])
"; done | tee -a $out/files_policy.txt
tail -3 $out/err.log
