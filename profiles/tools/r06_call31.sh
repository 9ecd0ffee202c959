#!/bin/bash
# Round 6: the light-page kernel beside the main one -- parity on the two new kernel forms first, then the in-process A/B.
export TMPDIR=/tmp
out=gpurun_out/r06c31; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_differential.py -m gpu -x -q -k "light3 or split3" 2>&1 | tail -15 | tee $out/tests.txt
BROTLIG_ENABLE_DEBUG_KNOBS=1 timeout 900 python profiles/tools/light_ab.py --workloads mixed runs records text --buckets 24 48 --reps 3 --steps 5 --out $out/light_ab.json 2>$out/err.log | tee $out/ab.txt
tail -5 $out/err.log
