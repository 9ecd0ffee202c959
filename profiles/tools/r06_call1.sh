#!/bin/bash
# Round 6, call 1: the GPU suite on the fused schedule kernel + job records, then an A/B against round 5's library (build/abv/lib_base.so).
export TMPDIR=/tmp
out=gpurun_out/r06c1; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $out/pytest.log | tail -3
timeout 600 python profiles/tools/ab_run.py --workloads mixed runs:1 text mixed:16:64 mixed:1:1 bc3 --reps 3 --steps 10 --out $out/ab.json 2>$out/ab.err | tee $out/ab.txt
tail -3 $out/ab.err
