#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06c2; mkdir -p $out; root=$(pwd)
for v in base r6d; do BROTLIG_HIP_SO=$root/build/abv/lib_$v.so timeout 300 python profiles/tools/r06_diag_status.py mixed 16 2>>$out/err.log; done | tee $out/diag.jsonl
timeout 600 python profiles/tools/ab_run.py --workloads mixed runs:1 mixed:16:64 bc3 --reps 3 --steps 10 --out $out/ab.json 2>>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
