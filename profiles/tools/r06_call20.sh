#!/bin/bash
# Round 6: a group's far-source loads issued before its flush and slide (BROTLIG_TUNE_EARLY_FAR=1): in-process A/B.
export TMPDIR=/tmp
out=gpurun_out/r06c20; mkdir -p $out
timeout 1400 python profiles/tools/ab_run.py --workloads mixed text files records samples16 runs bc3 --reps 3 --steps 5 --out $out/ab_earlyfar.json 2>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
