#!/bin/bash
# Round 6: does the device agree with the simulator that BROTLIG_TUNE_SHORT_COPY=40 is wrong?  (ab_run.py said "exact" for it.)
export TMPDIR=/tmp
out=gpurun_out/r06c48; mkdir -p $out
BROTLIG_HIP_SO=$(pwd)/build/abv/lib_short40.so timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -q -k "test_decode_gpu_plain" 2>&1 | tail -15 | tee $out/tests.txt
BROTLIG_HIP_SO=$(pwd)/build/abv/lib_short40.so timeout 600 python bench.py --workload runs --no-cpu-baseline --no-alt-parse 2>&1 | tail -3 | cut -c1-300
