#!/bin/bash
# Round 6, literal overflow through the queue (BROTLIG_TUNE_LIT_OVERFLOW=1): in-process A/B on every class.
export TMPDIR=/tmp
out=gpurun_out/r06c18; mkdir -p $out
timeout 1400 python profiles/tools/ab_run.py --workloads mixed text files bc3 records samples16 runs --reps 3 --steps 5 --out $out/ab_lito.json 2>$out/err.log | tee $out/ab.txt
tail -3 $out/err.log
