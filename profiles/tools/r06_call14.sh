#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06c14; mkdir -p $out
timeout 1500 python profiles/tools/ab_run.py --workloads files mixed text --reps 2 --steps 5 --out $out/ab_tunables.json 2>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
